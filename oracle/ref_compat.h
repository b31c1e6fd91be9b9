// TEST INFRASTRUCTURE ONLY -- force-included in front of the reference's headers when oracle/Makefile compiles the
// boundary proofs (ref_binding_main.cpp) against /root/reference/include: the reference is MSVC-flavoured and calls
// std::powf (Material.hpp:145), which libstdc++ 11 does not declare.  Nothing else is changed (no RNG swap here: the GPU
// binding never calls getRandomFloat).
#pragma once
#include <cmath>
#include <math.h>
namespace std { using ::powf; }

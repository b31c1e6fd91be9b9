// The framebuffer gather of the in-process N-device render (tutu_hip_render_multi_device) over RCCL: one communicator per
// distinct device (ncclCommInitAll, one process), ONE grouped ncclSend / ncclRecv exchange per frame into the first context's
// `gathered` buffer -- the C library's counterpart of tuturenderer_amd/dist.py (one process per GPU, torch.distributed) and of
// the reference's static row split over its threads (PathTracing.hpp:393-429).  RCCL is looked up at run time (dlopen
// librccl.so.1): the library has no link-time dependency on it, and where it is absent the gather stays on peer copies.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

namespace tutu {

struct RcclGather;  // communicators over a fixed list of distinct devices

// 1: librccl and the entry points this file uses were found (looked up once per process)
int rccl_available();
// communicators for `n` DISTINCT devices (ncclCommInitAll); nullptr + *err on failure
RcclGather* rccl_gather_create(const int* devices, int n, std::string* err);
void rccl_gather_destroy(RcclGather* g);
bool rccl_gather_matches(const RcclGather* g, const int* devices, int n);

struct RcclPiece {
	int device;              // where `src` lives (one of the communicator's devices)
	const float* src;        // the piece, on `device`
	float* dst;              // its place in the gathered buffer, on the root device
	size_t count;            // floats
	hipStream_t src_stream;  // a stream of `device` on which `src` is complete (or will be, in stream order)
};
// ONE ncclGroupStart ... ncclGroupEnd: every piece is sent by its device's rank and received by the root device's rank on
// root_stream (a piece that lives on the root device travels by a self send / recv).  Returns 0 or -1 (+ *err).
int rccl_gather_run(RcclGather* g, int root_device, hipStream_t root_stream, const RcclPiece* pieces, int n, std::string* err);

}  // namespace tutu

// rccl_gather.hpp: the RCCL leg of tutu_hip_render_multi_device.  Host code only.
#include "rccl_gather.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

namespace tutu {

namespace {

struct Api {
	void* lib = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
	bool ok = false;
};

const Api& api() {
	static Api a;
	static std::once_flag once;
	std::call_once(once, [] {
		// the soname first (a copy that is already in the process -- e.g. the one PyTorch ships -- is reused), then the ROCm tree
		const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
		for (const char* nm : names) {
			a.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
			if (a.lib) break;
		}
		if (!a.lib) return;
		a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(a.lib, "ncclCommInitAll"));
		a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
		a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(a.lib, "ncclGroupStart"));
		a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(a.lib, "ncclGroupEnd"));
		a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(a.lib, "ncclSend"));
		a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(a.lib, "ncclRecv"));
		a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
		a.ok = a.CommInitAll && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv && a.GetErrorString;
	});
	return a;
}

bool fail(std::string* err, const char* what, ncclResult_t r) {
	if (err) *err = std::string("RCCL: ") + what + ": " + (api().GetErrorString ? api().GetErrorString(r) : "?");
	return false;
}

}  // namespace

struct RcclGather {
	std::vector<int> devices;
	std::vector<ncclComm_t> comms;
};

int rccl_available() { return api().ok ? 1 : 0; }

RcclGather* rccl_gather_create(const int* devices, int n, std::string* err) {
	const Api& a = api();
	if (!a.ok) {
		if (err) *err = "RCCL: librccl.so.1 not found (or an entry point is missing)";
		return nullptr;
	}
	if (n <= 0) return nullptr;
	RcclGather* g = new RcclGather;
	g->devices.assign(devices, devices + n);
	g->comms.assign((size_t)n, nullptr);
	const ncclResult_t r = a.CommInitAll(g->comms.data(), n, g->devices.data());
	if (r != ncclSuccess) {
		fail(err, "ncclCommInitAll", r);
		delete g;
		return nullptr;
	}
	return g;
}

void rccl_gather_destroy(RcclGather* g) {
	if (!g) return;
	for (ncclComm_t c : g->comms)
		if (c) (void)api().CommDestroy(c);
	delete g;
}

bool rccl_gather_matches(const RcclGather* g, const int* devices, int n) {
	if (!g || (int)g->devices.size() != n) return false;
	for (int k = 0; k < n; k++)
		if (g->devices[(size_t)k] != devices[k]) return false;
	return true;
}

int rccl_gather_run(RcclGather* g, int root_device, hipStream_t root_stream, const RcclPiece* pieces, int n, std::string* err) {
	const Api& a = api();
	if (!g || !a.ok) return -1;
	auto rank_of = [&](int dev) {
		for (size_t k = 0; k < g->devices.size(); k++)
			if (g->devices[k] == dev) return (int)k;
		return -1;
	};
	const int root = rank_of(root_device);
	if (root < 0) {
		if (err) *err = "RCCL: the root device is not part of the communicator";
		return -1;
	}
	ncclResult_t r = a.GroupStart();
	if (r != ncclSuccess) {
		fail(err, "ncclGroupStart", r);
		return -1;
	}
	bool ok = true;
	for (int k = 0; k < n && ok; k++) {
		const RcclPiece& p = pieces[k];
		if (p.count == 0) continue;
		const int src = rank_of(p.device);
		if (src < 0) {
			if (err) *err = "RCCL: a piece lives on a device outside the communicator";
			ok = false;
			break;
		}
		// (a piece of the root device: a self send / recv, both on the root's stream)
		if (hipSetDevice(p.device) != hipSuccess) ok = false;
		if (ok && (r = a.Send(p.src, p.count, ncclFloat, root, g->comms[(size_t)src], src == root ? root_stream : p.src_stream)) != ncclSuccess) ok = fail(err, "ncclSend", r);
		if (ok && hipSetDevice(root_device) != hipSuccess) ok = false;
		if (ok && (r = a.Recv(p.dst, p.count, ncclFloat, src, g->comms[(size_t)root], root_stream)) != ncclSuccess) ok = fail(err, "ncclRecv", r);
	}
	r = a.GroupEnd();
	if (r != ncclSuccess && ok) ok = fail(err, "ncclGroupEnd", r);
	(void)hipSetDevice(root_device);
	return ok ? 0 : -1;
}

}  // namespace tutu

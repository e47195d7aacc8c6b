// A stand-in librccl.so.1 for tests/test_gpu_rccl_stub.py — TEST INFRASTRUCTURE, never shipped or linked.
//
// The engine moves its base-point table image between the GPUs of a group with the library's own RCCL calls (csrc/engine_group.inc:
// ncclCommInitAll, then one ncclBroadcast per rank between ncclGroupStart / ncclGroupEnd, then ncclCommDestroy), resolved with dlopen /
// dlsym — a path that needs two distinct GPUs and has therefore executed nowhere yet.  This stand-in exports the five symbols with the
// SIGNATURES OF RCCL's nccl.h (ROCm 7.2: /opt/rocm/include/rccl/rccl.h), appends every call with its arguments to the file named by
// KYB_RCCL_STUB_LOG, and performs the broadcast as device-to-device copies at ncclGroupEnd, so that on a one-GPU box the test can check
// (a) the argument values the library passes through its function-pointer types and (b) that the image really arrives.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>        // the real prototypes: a definition below that disagrees with them does not compile
#include <cstdio>
#include <cstdlib>
#include <vector>

struct ncclComm { int rank, ndev, device; };     // RCCL's ncclComm_t is a pointer to this opaque struct
typedef ncclComm StubComm;

extern "C" {
static FILE* stub_log() {
  static FILE* f = nullptr;
  if (!f) { const char* p = getenv("KYB_RCCL_STUB_LOG"); f = p ? fopen(p, "a") : stderr; if (!f) f = stderr; }
  return f;
}
struct Pending { const void* send; void* recv; size_t count; int dtype, root; ncclComm_t comm; hipStream_t stream; };
static std::vector<Pending> g_pending;
static int g_depth = 0;

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  fprintf(stub_log(), "ncclCommInitAll ndev=%d devlist=", ndev);
  for (int i = 0; i < ndev; ++i) fprintf(stub_log(), "%s%d", i ? "," : "", devlist ? devlist[i] : i);
  fprintf(stub_log(), "\n"); fflush(stub_log());
  const char* fail = getenv("KYB_RCCL_STUB_FAIL_INIT");
  if (fail) return (ncclResult_t)atoi(fail);
  for (int i = 0; i < ndev; ++i) comms[i] = new StubComm{i, ndev, devlist ? devlist[i] : i};
  return ncclSuccess;
}
ncclResult_t ncclGroupStart() { ++g_depth; fprintf(stub_log(), "ncclGroupStart\n"); fflush(stub_log()); return ncclSuccess; }
ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream) {
  int dev = -1;
  (void)hipGetDevice(&dev);
  fprintf(stub_log(), "ncclBroadcast rank=%d count=%zu dtype=%d root=%d in_place=%d in_group=%d current_device=%d comm_device=%d stream_null=%d\n", comm ? comm->rank : -1, count,
          (int)datatype, root, sendbuff == recvbuff, g_depth > 0, dev, comm ? comm->device : -1, stream == nullptr);
  fflush(stub_log());
  g_pending.push_back(Pending{sendbuff, recvbuff, count, (int)datatype, root, comm, stream});
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
  --g_depth;
  fprintf(stub_log(), "ncclGroupEnd pending=%zu\n", g_pending.size()); fflush(stub_log());
  const void* src = nullptr;
  for (const Pending& p : g_pending) if (p.comm->rank == p.root) src = p.send;
  int rc = src ? 0 : 5;
  for (const Pending& p : g_pending) {
    if (p.comm->rank == p.root || !src) continue;
    (void)hipSetDevice(p.comm->device);
    if (hipMemcpyAsync(p.recv, src, p.count, hipMemcpyDeviceToDevice, p.stream) != hipSuccess) rc = 1;
  }
  g_pending.clear();
  return (ncclResult_t)rc;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  fprintf(stub_log(), "ncclCommDestroy rank=%d\n", comm ? comm->rank : -1); fflush(stub_log());
  delete comm;
  return ncclSuccess;
}
}

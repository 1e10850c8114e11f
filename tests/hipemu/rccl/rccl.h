// TEST INFRASTRUCTURE — RCCL shim of the HIP-on-CPU emulator: the point-to-point subset the
// z-slab halo exchange uses.  Sends/receives posted inside a group are handed, at
// ncclGroupEnd, to a callback the test installs (tests drive it with torch.distributed/gloo,
// world_size 2), so the library's multi-rank host logic runs unmodified on CPU.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>

#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclCommEmu* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3,
               ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclFloat32 = 7, ncclFloat = 7,
               ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;

// kind: 1 = send, 0 = recv, 2 = float32 sum all-reduce (in place), 3 = float64 sum all-reduce
typedef struct { int kind; int peer; void* buf; size_t bytes; } hipemu_p2p_op;
typedef int (*hipemu_exchange_fn)(hipemu_p2p_op* ops, int n, void* user);
extern "C" void hipemu_set_exchange(hipemu_exchange_fn fn, void* user);

ncclResult_t ncclGetUniqueId(ncclUniqueId* id);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count);
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s);
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s);
ncclResult_t ncclAllReduce(const void* s, void* r, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t st);
const char* ncclGetErrorString(ncclResult_t r);

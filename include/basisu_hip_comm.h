/* include/basisu_hip_comm.h -- a native communicator for the sharded frontend (SURVEY.md 8e, 8b "comm_init(n_gpus)"): the two collectives of
 * bu_comm (include/basisu_hip_frontend.h) on RCCL over xGMI, enqueued on the context's stream, with no Python (or any host framework) in the
 * collective path. Lives in libbasisu_rccl.so (links librccl), so that libbasisu_hip.so itself does not depend on RCCL.
 *
 * One process per GPU:   rank 0 calls bu_rccl_get_unique_id and hands the 128 bytes to the other ranks by whatever channel the host
 *                        application has (MPI, torch.distributed's store, a file); every rank calls bu_rccl_comm_create.
 * One process, N GPUs:   bu_rccl_comm_init_all on N contexts (one per device), e.g. from the N threads of a C++ host such as the reference's
 *                        basis_parallel_compress (comp.cpp:5466).
 * Then bu_rccl_comm_fill gives the bu_comm for bu_frontend_set_comm. All int functions: 1 = success, 0 = failure (bu_rccl_last_error).
 */
#ifndef BASISU_HIP_COMM_H
#define BASISU_HIP_COMM_H
#include "basisu_hip_frontend.h"

#ifdef __cplusplus
extern "C" {
#endif

#define BU_RCCL_UNIQUE_ID_BYTES 128
typedef struct bu_rccl_comm bu_rccl_comm;

BU_HIP_API int bu_rccl_get_unique_id(void* out_id /* BU_RCCL_UNIQUE_ID_BYTES */);
BU_HIP_API bu_rccl_comm* bu_rccl_comm_create(bu_hip_context* ctx, const void* id, uint32_t rank, uint32_t world);
BU_HIP_API int bu_rccl_comm_init_all(bu_hip_context* const* ctxs, uint32_t n, bu_rccl_comm** out_comms /* n entries */);
BU_HIP_API void bu_rccl_comm_destroy(bu_rccl_comm*);
/* The bu_comm view of a communicator: all_gather = ncclAllGather in place over world * bytes_per_rank bytes, all_reduce_u64 = ncclAllReduce
 * (sum, uint64) in place; both ENQUEUED on the context's stream (bu_comm::stream_ordered = 1: no host synchronisation inside, the result is
 * ordered with everything enqueued on that stream afterwards). `out->user` points at the communicator.
 * Threading: drive each communicator from its own host thread (one thread per GPU, as basis_parallel_compress does); the N communicators of
 * bu_rccl_comm_init_all must not be driven in turn from ONE thread -- a collective only completes once every rank has enqueued its part. */
BU_HIP_API int bu_rccl_comm_fill(bu_rccl_comm*, bu_comm* out);
/* The threading rule above is ENFORCED for the communicators of one bu_rccl_comm_init_all call, on the sequence of collectives the ranks share: a host thread that has
 * issued collective number e (or a later one) for one rank and then issues number e for ANOTHER rank of the group gets 0 (bu_rccl_last_error names both ranks and the
 * number) instead of waiting for ever. Nothing else is bound: a rank's next collective may come from any thread (executor pools), and threads are told apart by tokens
 * that are never reused, not by std::thread::id.
 * Test hook: n communicators of one group with no RCCL communicator behind them (their collectives fail with "no communicator" once the rule has let them through). */
BU_HIP_API int bu_rccl_debug_unconnected_group(uint32_t n, bu_rccl_comm** out_comms);
BU_HIP_API const char* bu_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif

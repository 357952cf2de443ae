// One-shot all-reduce over peer-mapped mailboxes (include/aaerec_hip.h: aae_ipc_*) - an aae_collectives table for the ranks of
// ONE node, for the small latency-bound exchanges of aae_shard_step (three all-reduces of [global rows, hidden] partial sums
// per step: 0.16 - 0.65 MB at 2 - 8 ranks).  A ring collective moves such a buffer in 2 (world - 1) dependent hops over the
// point-to-point xGMI links; here every rank
//     1. copies its operand into ITS OWN mailbox (device memory exported with hipIpcGetMemHandle, mapped by every peer),
//        and publishes the call's number in the mailbox's flag word (system-scope release),
//     2. waits until every peer has published that number (system-scope acquire loads of the peers' flag words),
//     3. adds the world mailboxes in RANK ORDER straight into its operand (loads that bypass the non-coherent caches):
// one launch, one hop, world - 1 remote reads of the buffer per rank, and the same bits on every rank (a fixed order of
// addition - what keeps the replicated hidden layers of the `shard` scheme identical without a gradient exchange).
// Two slots per mailbox, alternating by call: a rank can be at most one call ahead of a peer that still reads (it publishes
// call k + 1 only after it has seen every peer's k, and enters k + 2 - the slot of k again - only after every peer published
// k + 1, i.e. finished reading k).
// The wait SPINS on the device: every rank's launch has to be resident at the same time.  That holds for one process per GPU
// (the data-parallel launch of this repository) and, on the single-GPU boxes this was built on, for two processes sharing the
// card (each has hardware queues of its own; the launch is 32 workgroups of 256 threads).  The spin is BOUNDED (kIpcSpinTicks of
// the 100 MHz clock, ~2 s): a rank that never sees its peers raises the mailbox's error word and goes on - the host reads it
// at the table's next call / at aae_ipc_destroy and fails loudly instead of hanging the device.
// No counterpart in the reference (it has no distributed code); the exchange it serves is DESIGN.md 5's `shard` scheme.
#pragma once
#include "device_common.h"

namespace aae {

constexpr int kIpcMaxWorld = 16;
constexpr int kIpcBlocks = 32;
constexpr unsigned long long kIpcSpinTicks = 200000000ull;     // 2 s of the 100 MHz clock
constexpr size_t kIpcHeaderBytes = 256;                          // flag word, arrival counter, error word (+ padding)

struct IpcPeers {
    const float* slot[kIpcMaxWorld];        // mailbox data of rank r (this call's slot), mapped here
    const unsigned* flag[kIpcMaxWorld];     // ... its flag word
};

// header words of a mailbox: [0] flag (the number of the last published call), [1] arrival counter of the local workgroups,
// [2] error word (a peer's flag was not seen in time), [3] departure counter
__global__ __launch_bounds__(256) void ipc_allreduce_kernel(float* __restrict__ buf, long long count, float* own_slot, unsigned* own_hdr,
                                                           IpcPeers peers, int world, int rank, unsigned call) {
    const int tid = threadIdx.x;
    const long long n4 = count >> 2;        // (count is a multiple of 4: the activations' padded row stride)
    const long long per = (n4 + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    // 1. operand -> own mailbox; the last workgroup to finish publishes the call
    for (long long i = lo + tid; i < hi; i += 256) reinterpret_cast<float4*>(own_slot)[i] = reinterpret_cast<const float4*>(buf)[i];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const unsigned arrived = __hip_atomic_fetch_add(own_hdr + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == gridDim.x * call - 1u)       // (the counter runs on: call k ends at k x blocks)
            __hip_atomic_store(own_hdr, call, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. every peer's flag
    if (tid < world && tid != rank) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(peers.flag[tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < call) {
            if (wall_clock64() - t0 > kIpcSpinTicks) { __hip_atomic_store(own_hdr + 2, call, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    // 3. the sum in rank order (system-scope loads: a peer's mailbox is not coherent in this device's caches)
    for (long long i = lo + tid; i < hi; i += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < world; ++r) {
            const float* p = peers.slot[r] + 4 * i;
            float4 v;
            v.x = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            v.z = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            v.w = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (r == 0) acc = v;
            else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
        reinterpret_cast<float4*>(buf)[i] = acc;
    }
}

struct IpcCtx {
    int world, rank;
    size_t cap_floats;                      // floats per slot
    char* base[kIpcMaxWorld];               // mailbox of rank r as mapped in this process (own: the allocation itself)
    bool opened[kIpcMaxWorld];
    unsigned call;
};

inline float* ipc_slot(char* base, size_t cap, unsigned call) { return reinterpret_cast<float*>(base + kIpcHeaderBytes) + (size_t)(call & 1u) * cap; }

static int ipc_check_error(IpcCtx* c, hipStream_t s) {
    unsigned err = 0;
    if (hipMemcpyAsync(&err, c->base[c->rank] + 8, sizeof(err), hipMemcpyDeviceToHost, s) != hipSuccess) return fail(AAE_EHIP, "aae_ipc: reading the mailbox's error word failed");
    if (hipStreamSynchronize(s) != hipSuccess) return fail(AAE_EHIP, "aae_ipc: synchronising for the error word failed");
    if (err) return fail(AAE_ESTATE, "aae_ipc all_reduce: a peer did not publish call " + std::to_string(err) + " within 2 s (is every rank's launch resident?)");
    return AAE_OK;
}

static int ipc_all_reduce(void* ctx, float* buf, int64_t count, void* stream) {
    IpcCtx* c = static_cast<IpcCtx*>(ctx);
    if (count <= 0) return AAE_OK;
    if (count & 3) return fail(AAE_EINVAL, "aae_ipc all_reduce: count must be a multiple of 4 floats");
    if ((size_t)count > c->cap_floats) return fail(AAE_EINVAL, "aae_ipc all_reduce: the operand exceeds the mailboxes' capacity (aae_ipc_create)");
    if (c->world == 1) return AAE_OK;
    const unsigned call = ++c->call;
    IpcPeers p;
    for (int r = 0; r < c->world; ++r) {
        p.slot[r] = ipc_slot(c->base[r], c->cap_floats, call);
        p.flag[r] = reinterpret_cast<const unsigned*>(c->base[r]);
    }
    hipLaunchKernelGGL(ipc_allreduce_kernel, dim3(kIpcBlocks), dim3(256), 0, S(stream), buf, (long long)count,
                       ipc_slot(c->base[c->rank], c->cap_floats, call), reinterpret_cast<unsigned*>(c->base[c->rank]), p, c->world, c->rank, call);
    LAUNCHCHK("ipc_allreduce");
    if ((call & 1023u) == 0) return ipc_check_error(c, S(stream));      // (a periodic look at the error word: one sync per 1024 calls)
    return AAE_OK;
}
// all_gather / reduce_scatter through the same mailboxes are not built: aae_shard_step, the scheme this table serves, calls
// all_reduce only (aae_dp_step's seven collectives move megabytes - RCCL's job)
static int ipc_unsupported(const char* what) { return fail(AAE_ESTATE, std::string("aae_ipc collectives: ") + what + " is not available (all_reduce only: aae_shard_step)"); }
static int ipc_all_gather(void*, const float*, float*, int64_t, void*) { return ipc_unsupported("all_gather"); }
static int ipc_reduce_scatter(void*, const float*, float*, int64_t, void*) { return ipc_unsupported("reduce_scatter"); }

}  // namespace aae

extern "C" {

int aae_ipc_create(int64_t max_floats, char handle_out[64], void** mailbox_out) {
    if (max_floats < 4 || !handle_out || !mailbox_out) return fail(AAE_EINVAL, "aae_ipc_create: max_floats >= 4, handle_out, mailbox_out");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the 64 bytes of handle_out");
    const size_t cap = ((size_t)max_floats + 3) & ~(size_t)3;
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, kIpcHeaderBytes + 2 * cap * sizeof(float)));
    HIPCHK(hipMemset(p, 0, kIpcHeaderBytes + 2 * cap * sizeof(float)));
    HIPCHK(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, p) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return fail(AAE_EHIP, "hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)"); }
    memcpy(handle_out, &h, 64);
    *mailbox_out = p;
    return AAE_OK;
}

int aae_ipc_init(void* mailbox, const char* handles, int32_t world, int32_t rank, int64_t max_floats, aae_collectives* out) {
    if (!mailbox || !handles || !out) return fail(AAE_EINVAL, "aae_ipc_init: NULL argument");
    if (world < 1 || world > kIpcMaxWorld || rank < 0 || rank >= world) return fail(AAE_EINVAL, "aae_ipc_init: need 0 <= rank < world <= 16");
    IpcCtx* c = new IpcCtx();
    memset((void*)c, 0, sizeof(*c));
    c->world = world; c->rank = rank; c->cap_floats = ((size_t)max_floats + 3) & ~(size_t)3; c->call = 0;
    for (int r = 0; r < world; ++r) {
        if (r == rank) { c->base[r] = static_cast<char*>(mailbox); continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)r * 64, 64);
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            for (int q = 0; q < r; ++q) if (c->opened[q]) (void)hipIpcCloseMemHandle(c->base[q]);
            delete c;
            return fail(AAE_EHIP, "hipIpcOpenMemHandle failed for rank " + std::to_string(r) + "'s mailbox");
        }
        c->base[r] = static_cast<char*>(p); c->opened[r] = true;
    }
    out->ctx = c; out->world = world; out->rank = rank;
    out->all_gather = ipc_all_gather; out->reduce_scatter = ipc_reduce_scatter; out->all_reduce = ipc_all_reduce;
    return AAE_OK;
}

int aae_ipc_destroy(aae_collectives* c, void* mailbox) {
    int rc = AAE_OK;
    if (c && c->ctx) {
        IpcCtx* x = static_cast<IpcCtx*>(c->ctx);
        (void)hipDeviceSynchronize();
        rc = ipc_check_error(x, nullptr);
        for (int r = 0; r < x->world; ++r) if (x->opened[r]) (void)hipIpcCloseMemHandle(x->base[r]);
        delete x;
        c->ctx = nullptr;
    }
    if (mailbox) (void)hipFree(mailbox);
    return rc;
}

}  // extern "C"

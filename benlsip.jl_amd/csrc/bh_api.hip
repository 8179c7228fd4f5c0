// bh_api.hip — C ABI (include/benlsip_hip.h) over the gfx950 kernels of bh_kernels.hip.h.
// One process drives one GPU; all launches go to one stream; exports that move host data are synchronous, device-pointer
// exports return once the host has what it is owed (option final_sync).
#include "../../include/benlsip_hip.h"
#include "bh_kernels.hip.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>   // types only: librccl is dlopen'ed in bh_comm_init (never needed on one GPU)
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <string>
#include <thread>
#include <vector>

using namespace bh;

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
namespace {

struct CgWorkspace {
    int64_t n_pad = 0;
    double *w = nullptr, *r = nullptr, *v = nullptr, *p = nullptr, *Hp = nullptr, *g = nullptr, *wl = nullptr, *wu = nullptr;
    double *x = nullptr, *s = nullptr, *xlow = nullptr, *xupp = nullptr;   // minor_iterate staging
    double* hw = nullptr;          // H*w accumulated by the CG loop for minor_iterate's linesearch
    double *p2 = nullptr, *gpart = nullptr, *rvpart = nullptr;   // two-kernel box iteration: p ping-pong, (spare), r.v partials (2 x n_pad/2)
    double *r2 = nullptr, *hpx = nullptr, *hpx_tail = nullptr;   // RCCL form of it: r ping-pong; H*p with the p'Hp slot right behind it (hpx[n_pad] = hpx_tail[0])
    double* slab = nullptr;
    double* scalars = nullptr;     // 8 doubles (linesearch alpha, ...)
    CgState* d_state = nullptr;
    volatile unsigned long long* h_mirror = nullptr;   // host-mapped progress word written by the CG kernels
    unsigned long long* d_mirror = nullptr;            // its device address
    unsigned tag = 0;
    double* d_trace = nullptr;
    int64_t trace_cap = 0;
};

// Host side of the peer-buffer exchange: this rank's inbox, the peers' inboxes mapped through hipIpc, and the POSIX
// shared-memory page the ranks of one node meet on to swap handles (named after the communicator's unique id).
struct PeerShm {
    std::atomic<int> arrive;
    std::atomic<int> generation;
    int failed;
    hipIpcMemHandle_t handle[kMaxPeers];
};
struct PeerComm {
    bool active = false;
    bool broken = false;                     // an exchange timed out: the path may not be selected again
    void* inbox = nullptr;                   // slots | flags | seq, arrive   (one allocation: one IPC handle)
    size_t inbox_bytes = 0;
    void* peer_base[kMaxPeers] = {};         // mapped inboxes (own rank: inbox)
    volatile unsigned long long* h_err = nullptr;   // host-mapped error word
    PeerShm* shm = nullptr;
    PeerArgs args{};
};
constexpr int64_t kPeerCap = 1 << 16;                                   // doubles per slot (512 KiB): larger vectors go in pieces
constexpr int kPeerBlkCap = (int)(kPeerCap / (2 * kPeerBlockChunks));   // 2048 flags per (parity, rank)
constexpr size_t kPeerSlotBytes = (size_t)2 * kMaxPeers * kPeerCap * sizeof(double);
constexpr size_t kPeerFlagBytes = (size_t)2 * kMaxPeers * kPeerBlkCap * sizeof(unsigned long long);
constexpr size_t kPeerScalBytes = kPeerFlagBytes;                         // one double per (parity, rank, block): scalars that ride along

// An upload in flight (bh_hess_create_async): a worker thread feeds column chunks of the caller's J through two device
// staging buffers — copy of chunk k+1 on one stream while chunk k is transposed into the row-major image on another.
struct AsyncUpload {
    std::thread worker;
    int32_t rc = BH_OK;              // written by the worker, read after join
    std::string detail;
    hipStream_t s_copy = nullptr, s_xpose = nullptr;
    double* staging[2] = {nullptr, nullptr};
    hipEvent_t copied[2] = {nullptr, nullptr}, freed[2] = {nullptr, nullptr};
    size_t staging_bytes = 0;        // capacity of each staging buffer
    int64_t chunk_cols = 0;
    bool resources() const { return s_copy != nullptr; }
};

struct Ctx {
    bool init = false;
    int device = -1;
    int flags = 0;
    int n_cu = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string detail;
    // options
    int64_t opt_blocks_per_cu = 0;   // 0 = per-config default
    int64_t opt_variant = 0;         // kernel geometry variant for the 2048-chunk (n<=4096) class
    int64_t opt_batch = 0;           // CG iterations launched ahead of the host's done-flag poll; 0 = by problem size (auto_batch)
    int64_t opt_chol_blocked = 1;    // mA > 64: blocked potrf/trsm/syrk chain (0: one-workgroup right-looking kernel)
    // Cauchy search, per breakpoint: 0 = downdate the Gram matrix A_free A_free' and refactor it (O(mA^3), the default: its
    // factor is as accurate as the reference's from-scratch rebuild), 1 = for mA > 64: rank-one downdate of the factor itself
    // (O(mA^2)), refreshed from scratch every kDowndateRefresh breakpoints.  Errors accumulate over the downdates — measured on the
    // 48-parameter NLS (mA = 2, when the one-wave kernel for mA <= 64 still existed): after 29 of them a search direction P(-g)
    // with ||g||/||P(-g)|| = 1.7e7 took one more breakpoint than the oracle; after 40, with A_free A_free' close to singular, the
    // projections had left null(A).  Up to 64 rows the refactoring path costs the same, so it is the only one there.
    int64_t opt_chol_downdate = 0;
    // bh_cauchy_step with box constraints on one rank: 1 = image-space search (J d and J s_c maintained by rank-one column updates:
    // one J v sweep at the start, then no sweep over J per breakpoint), 0 = one H*d sweep per breakpoint as the reference does
    int64_t opt_cauchy_image = 1;
    int64_t opt_cauchy_image_max_ma = 64;   // ... and with up to this many linear equalities (0..64)
    int64_t opt_cauchy_fused = 1;           // box constraints, one rank, row-space form: ONE kernel per breakpoint (cauchy_fused_kernel)
    int64_t opt_cauchy_fused_grid = 0;      // experiment: workgroups of cauchy_fused_kernel (0: one row per thread up to kCauchyFusedGrid)
    int64_t opt_linv_refine = 1;            // explicit-inverse projection (three-kernel CG iteration): one step of iterative refinement of y
    int64_t opt_cauchy_gemm = 1;            // B = J D A' of that form in one sweep on the matrix cores (0: mA J v sweeps over masked rows of A)
    int64_t opt_gram_mfma = 1;       // A_free A_free': 1 = matrix cores when mA > 96, 2 = always, 0 = never (one wave per entry, VALU)
    int64_t opt_ls_from_cg = 1;      // minor_iterate: linesearch's w'Hw from the H*w accumulated by the CG loop
    // bh_step_accumulate_dev right behind the bh_minor_iterate_dev that produced its w: g_minor += H*w with the H*w that CG loop
    // accumulated (scaled with w by the line search) instead of a fresh sweep H*s + g over J — one H-product less per minor iterate.
    // Opt-in: it relies on the CALLER'S invariant that g_minor_out holds H*s + g for the current s (true in inner_step,
    // src/basic_tralcnlss.jl:412,:434-437; the library's resident inner-step mirrors switch it on around their loop)
    int64_t opt_step_from_cg = 0;
    struct { const void* H = nullptr; const double* w = nullptr; const double* gm = nullptr; } hw_note;   // what cg.hw currently is H*w of
    // the last (H, s, g, g_minor) for which the library itself wrote g_minor = H*s + g (bh_hmul_add_dev, bh_step_accumulate_dev):
    // with step_from_cg on, bh_model_reduction_dev on the same H, s, g takes s'Hs = s.(g_minor - g) instead of sweeping J
    struct { const void* H = nullptr; const double* s = nullptr; const double* g = nullptr; const double* gm = nullptr; } gm_note;
    int64_t opt_fold_init = 1;       // box CG: fold the initialisation into the first H*p / step launches
    int64_t opt_final_sync = 0;      // *_dev: always drain the stream before returning (1), or only wait for what the host is owed (0)
    // the end-of-call wait of the host-pointer entry points: hipStreamSynchronize (0) or a mailbox seal + poll (1: A/B'd, SLOWER —
    // a kernel behind a D2H DMA pays a cross-engine dependency: bh_pcg 0.667 -> 0.72 ms, bh_project 34 -> 45-50 us; kept as a switch)
    int64_t opt_mbox_flush = 0;
    int64_t opt_host_copy_kernels = 1;   // host vectors of the host-pointer entry points: copy kernels on the mapped pinned arena (1) or DMA (0)
    int64_t opt_cg_fused = 1;        // box CG: two kernels per iteration (H*p with the p-update folded in + reduce/update) instead of three
    int64_t opt_proj_form = 1;       // 1: reduced mA x mA form (fast), 0: the reference's augmented mpp x mpp form
    int64_t opt_upload_chunk_mb = 64; // bh_hess_create_async: MiB of J per pipelined column chunk
    int64_t opt_ev_stride = 8;       // BH_FLAG_PROFILE: hipEvents around every opt_ev_stride-th H*p launch of a handle
    int64_t opt_pingpong = 0;        // alternate the sweep direction of J between consecutive H*p products (A/B: +1 % without nt loads, -0.2 % with)
    // RCCL
    void* rccl_lib = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    ncclResult_t (*p_ncclGetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*p_ncclCommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*p_ncclCommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*p_ncclAllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*p_ncclGetErrorString)(ncclResult_t) = nullptr;
    // one-shot all-reduce over peer-mapped inboxes (bh_comm.hip.h); comm_path selects it (1) or RCCL (0) when both exist
    PeerComm peer;
    int comm_path = 0;
    CgWorkspace cg;
    double* scratch_dev = nullptr;   // small device scratch (selftest, f2b)
    // Jacobian images of destroyed handles, kept for the next bh_hess_create* of a similar size: the reference builds a new
    // AlHessian for every accepted step (src/basic_tralcnlss.jl:361-362) and drops the old one — a 2 GiB hipFree makes the
    // driver scrub the memory in the background (-5 % HBM bandwidth for ~60 ms) and the next hipMalloc costs ~4 ms
    struct PooledImage { double* ptr; int64_t doubles; };
    std::vector<PooledImage> image_pool;
    int64_t opt_image_pool = 2;      // images kept (0: free at once)
    int64_t image_pool_hits = 0;
    AsyncUpload* upload_cache = nullptr;   // streams, events and staging buffers of the last finished asynchronous upload, kept for the next
    // host <-> device traffic issued by the library since bh_init (bh_stats: the device-resident entry points are checked against it)
    int64_t h2d_bytes = 0, d2h_bytes = 0, h2d_calls = 0, d2h_calls = 0;
    // Mailbox: a host-mapped page the kernels write the few host-visible results of a device-pointer call into (counts, norms,
    // alpha, the BitVector image), sealed by a sequence number: the host polls one word instead of paying for a DMA per
    // scalar (~10 us each) and a hipStreamSynchronize (~10 us) per call.
    volatile unsigned long long* mbox_h = nullptr;   // [0] = sequence number, payload from byte 64
    unsigned long long* mbox_d = nullptr;
    unsigned long long mbox_seq = 0;
    double* rbuf = nullptr;          // residual staging of bh_resid_sqnorm (grown on demand)
    int64_t rbuf_cap = 0;
    int live_hess = 0;               // bh_hess handles alive (a handle bakes in this rank's share of C: see bh_comm_init)
    // dynamic-LDS ceilings already raised on this device (hipFuncSetAttribute); reset by bh_shutdown
    bool vlds_attr_set = false;      // row_stream_kernel<512,16,1,FUSED,...,VL>
    bool cgp_vlds_attr_set = false;  // ... and its two CG-prologue symbols
    bool trsm_lds_granted = false;   // chol_trsm_kernel
    size_t trsv_lds_granted = 0;     // trsv_pair_kernel
};

Ctx g_ctx;

void pin_arena_abandon(bool drained);   // drops results parked for a call that is about to fail
bool pin_arena_busy();

// drain: DMAs queued by stage_vec / fetch_vec may still be reading or writing the pinned arena; wait for them before the
// arena is handed to the next call.  A caller that is failing BECAUSE the stream makes no progress passes drain = false
// (the arena is then retired instead of reused).
int32_t fail(int32_t code, const std::string& what, bool drain = true) {
    g_ctx.detail = what;
    // the runtime keeps the last error until somebody reads it: a failed hipMalloc (out of memory) would otherwise surface
    // again in the next call's hipGetLastError() check and fail a perfectly good launch
    (void)hipGetLastError();
    bool drained = true;
    if (pin_arena_busy()) drained = drain && g_ctx.stream != nullptr && hipStreamSynchronize(g_ctx.stream) == hipSuccess;
    pin_arena_abandon(drained);
    return code;
}

#define BH_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(BH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));               \
    } while (0)

#define BH_NCCL(call)                                                                                  \
    do {                                                                                               \
        ncclResult_t r__ = (call);                                                                     \
        if (r__ != ncclSuccess)                                                                        \
            return fail(BH_ERR_RCCL, std::string(#call) + ": " +                                      \
                                         (g_ctx.p_ncclGetErrorString ? g_ctx.p_ncclGetErrorString(r__) : "?")); \
    } while (0)

#define BH_REQUIRE_INIT() \
    do { if (!g_ctx.init) return fail(BH_ERR_NOT_INIT, "bh_init has not been called"); } while (0)

#define BH_TRY(expr) do { int32_t rc__ = (expr); if (rc__ != BH_OK) return rc__; } while (0)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline void count_h2d(size_t bytes) { g_ctx.h2d_bytes += (int64_t)bytes; g_ctx.h2d_calls += 1; }
inline void count_d2h(size_t bytes) { g_ctx.d2h_bytes += (int64_t)bytes; g_ctx.d2h_calls += 1; }

template <class T>
int32_t dev_alloc(T** out, int64_t count) {
    *out = nullptr;
    if (count <= 0) count = 1;
    BH_HIP(hipMalloc(reinterpret_cast<void**>(out), (size_t)count * sizeof(T)));
    return BH_OK;
}

void dev_free(void* p) { if (p) (void)hipFree(p); }

// ------------------------------------------------------------------------------------------
// row-stream kernel geometry
// ------------------------------------------------------------------------------------------
struct RsConfig { int T, CPT, R, blocks_per_cu; };

// index -> template instantiation (see launch_row_stream)
const RsConfig kRsConfigs[] = {
    {64, 1, 8, 8},     // 0: nchunks <= 64    (n <= 128)
    {256, 1, 8, 4},    // 1: nchunks <= 256   (n <= 512)
    {256, 2, 8, 2},    // 2: nchunks <= 512   (n <= 1024)   2 WGs/CU measured best (6.5 vs 6.0 TB/s at 4)
    {256, 4, 4, 2},    // 3: nchunks <= 1024  (n <= 2048)   2 WGs/CU measured best (6.6 vs 6.0 TB/s at 4)
    {256, 8, 4, 1},    // 4: nchunks <= 2048  (n <= 4096)   variant 0 (measured best: 1 WG/CU, 128 KiB in flight)
    {512, 8, 2, 1},    // 5: nchunks <= 4096  (n <= 8192)
    {512, 4, 4, 1},    // 6: nchunks <= 2048  variant 1
    {256, 8, 2, 1},    // 7: nchunks <= 2048  variant 2
    {1024, 2, 4, 1},   // 8: nchunks <= 2048  variant 3
    {512, 4, 2, 1},    // 9: nchunks <= 2048  variant 4
    {256, 8, 4, 1},    // 10: nchunks <= 2048 variant 5 = variant 0 WITHOUT non-temporal loads of J (A/B: nt = +10 %)
    {256, 8, 4, 2},    // 11: variant 6 = one register buffer (no prefetch), 2 WGs/CU
    {256, 8, 3, 1},    // 12: variant 7 = R = 3 with prefetch
    {256, 8, 6, 1},    // 13: variant 8 = R = 6, one register buffer
    {512, 16, 1, 1},   // 14: nchunks <= 8192 (n <= 16384): one row per step; the fused mode parks v in LDS (VL)
};
constexpr int64_t kMaxChunks = 8192;
constexpr int64_t kMaxBlocksPerCu = 8;   // partial-slab capacity per handle: n_cu * kMaxBlocksPerCu workgroups (alloc_hess_common)

int pick_config(int nchunks) {
    if (nchunks <= 64) return 0;
    if (nchunks <= 256) return 1;
    if (nchunks <= 512) return 2;
    if (nchunks <= 1024) return 3;
    if (nchunks <= 2048) {
        switch (g_ctx.opt_variant) {
            case 1: return 6;
            case 2: return 7;
            case 3: return 8;
            case 4: return 9;
            case 5: return 10;
            case 6: return 11;
            case 7: return 12;
            case 8: return 13;
            default: return 4;
        }
    }
    if (nchunks <= 4096) return 5;
    return 14;
}

template <int T, int CPT, int R, int NT = 1, int PF = 1>
void launch_rs_mode(int mode, const RowStreamArgs& a, int grid, hipStream_t s) {
    switch (mode) {
        case MODE_JV: hipLaunchKernelGGL((row_stream_kernel<T, CPT, R, MODE_JV, NT, PF>), dim3(grid), dim3(T), 0, s, a); break;
        case MODE_JTV: hipLaunchKernelGGL((row_stream_kernel<T, CPT, R, MODE_JTV, NT, PF>), dim3(grid), dim3(T), 0, s, a); break;
        default: hipLaunchKernelGGL((row_stream_kernel<T, CPT, R, MODE_FUSED, NT, PF>), dim3(grid), dim3(T), 0, s, a); break;
    }
}

// MODE_FUSED with the CG prologue (two-kernel box iteration); default geometries only.  CGP = 2: the launch expected to stop.
template <int T, int CPT, int R>
void launch_rs_cgp(const RowStreamArgs& a, int grid, hipStream_t s, bool expect_stop) {
    if (expect_stop) hipLaunchKernelGGL((row_stream_kernel<T, CPT, R, MODE_FUSED, 1, 1, 0, 2>), dim3(grid), dim3(T), 0, s, a);
    else hipLaunchKernelGGL((row_stream_kernel<T, CPT, R, MODE_FUSED, 1, 1, 0, 1>), dim3(grid), dim3(T), 0, s, a);
}

// n <= 16384: J v and J'u keep everything in registers; the fused mode needs the v slice in LDS (T * CPT * 16 bytes = 128 KiB,
// above the 64 KiB a kernel may use without asking).
template <int T, int CPT, int R>
void launch_rs_mode_vlds(int mode, const RowStreamArgs& a, int grid, hipStream_t s) {
    if (mode != MODE_FUSED) { launch_rs_mode<T, CPT, R>(mode, a, grid, s); return; }
    constexpr size_t lds = (size_t)T * CPT * sizeof(double2);
    if (!g_ctx.vlds_attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&row_stream_kernel<T, CPT, R, MODE_FUSED, 1, 1, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        g_ctx.vlds_attr_set = true;
    }
    hipLaunchKernelGGL((row_stream_kernel<T, CPT, R, MODE_FUSED, 1, 1, 1>), dim3(grid), dim3(T), lds, s, a);
}

void launch_row_stream(int cfg, int mode, const RowStreamArgs& a, int grid, hipStream_t s) {
    switch (cfg) {
        case 0: launch_rs_mode<64, 1, 8>(mode, a, grid, s); break;
        case 1: launch_rs_mode<256, 1, 8>(mode, a, grid, s); break;
        case 2: launch_rs_mode<256, 2, 8>(mode, a, grid, s); break;
        case 3: launch_rs_mode<256, 4, 4>(mode, a, grid, s); break;
        case 4: launch_rs_mode<256, 8, 4>(mode, a, grid, s); break;
        case 5: launch_rs_mode<512, 8, 2>(mode, a, grid, s); break;
        case 6: launch_rs_mode<512, 4, 4>(mode, a, grid, s); break;
        case 7: launch_rs_mode<256, 8, 2>(mode, a, grid, s); break;
        case 8: launch_rs_mode<1024, 2, 4>(mode, a, grid, s); break;
        case 10: launch_rs_mode<256, 8, 4, 0>(mode, a, grid, s); break;
        case 11: launch_rs_mode<256, 8, 4, 1, 0>(mode, a, grid, s); break;
        case 12: launch_rs_mode<256, 8, 3, 1, 1>(mode, a, grid, s); break;
        case 13: launch_rs_mode<256, 8, 6, 1, 0>(mode, a, grid, s); break;
        case 14: launch_rs_mode_vlds<512, 16, 1>(mode, a, grid, s); break;
        default: launch_rs_mode<512, 4, 2>(mode, a, grid, s); break;
    }
}

// CGP = 3: the RCCL form (update of the previous iteration folded into the prologue); register-resident geometries only.
bool cgp3_supported(int cfg) { return cfg <= 5; }
void launch_row_stream_cgp3(int cfg, const RowStreamArgs& a, int grid, hipStream_t s) {
    switch (cfg) {
        case 0: hipLaunchKernelGGL((row_stream_kernel<64, 1, 8, MODE_FUSED, 1, 1, 0, 3>), dim3(grid), dim3(64), 0, s, a); break;
        case 1: hipLaunchKernelGGL((row_stream_kernel<256, 1, 8, MODE_FUSED, 1, 1, 0, 3>), dim3(grid), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((row_stream_kernel<256, 2, 8, MODE_FUSED, 1, 1, 0, 3>), dim3(grid), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((row_stream_kernel<256, 4, 4, MODE_FUSED, 1, 1, 0, 3>), dim3(grid), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((row_stream_kernel<256, 8, 4, MODE_FUSED, 1, 1, 0, 3>), dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((row_stream_kernel<512, 8, 2, MODE_FUSED, 1, 1, 0, 3>), dim3(grid), dim3(512), 0, s, a); break;
    }
}

bool cgp_supported(int cfg) { return cfg <= 5 || cfg == 14; }
void launch_row_stream_cgp(int cfg, const RowStreamArgs& a, int grid, hipStream_t s, bool expect_stop) {
    switch (cfg) {
        case 0: launch_rs_cgp<64, 1, 8>(a, grid, s, expect_stop); break;
        case 1: launch_rs_cgp<256, 1, 8>(a, grid, s, expect_stop); break;
        case 2: launch_rs_cgp<256, 2, 8>(a, grid, s, expect_stop); break;
        case 3: launch_rs_cgp<256, 4, 4>(a, grid, s, expect_stop); break;
        case 4: launch_rs_cgp<256, 8, 4>(a, grid, s, expect_stop); break;
        case 5: launch_rs_cgp<512, 8, 2>(a, grid, s, expect_stop); break;
        default: {
            constexpr size_t lds = (size_t)512 * 16 * sizeof(double2);
            if (!g_ctx.cgp_vlds_attr_set) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&row_stream_kernel<512, 16, 1, MODE_FUSED, 1, 1, 1, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&row_stream_kernel<512, 16, 1, MODE_FUSED, 1, 1, 1, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                g_ctx.cgp_vlds_attr_set = true;
            }
            if (expect_stop) hipLaunchKernelGGL((row_stream_kernel<512, 16, 1, MODE_FUSED, 1, 1, 1, 2>), dim3(grid), dim3(512), lds, s, a);
            else hipLaunchKernelGGL((row_stream_kernel<512, 16, 1, MODE_FUSED, 1, 1, 1, 1>), dim3(grid), dim3(512), lds, s, a);
            break;
        }
    }
}

int grid_for(int cfg, int64_t nrows) {
    const RsConfig& c = kRsConfigs[cfg];
    const int64_t ngroups = (nrows + c.R - 1) / c.R;
    const int64_t bpc = g_ctx.opt_blocks_per_cu > 0 ? g_ctx.opt_blocks_per_cu : c.blocks_per_cu;
    int64_t g = (int64_t)g_ctx.n_cu * std::min<int64_t>(bpc, kMaxBlocksPerCu);   // the slab buffers hold n_cu * kMaxBlocksPerCu rows
    g = std::min<int64_t>(g, std::max<int64_t>(ngroups, 1));
    return (int)g;
}

constexpr int kEvCap = 512;
constexpr int kCauchyFusedGrid = 256;    // workgroups of cauchy_fused_kernel (512 rows each per sweep); its partial sums are 2 x [2][256]
constexpr int kCauchyImgGrid = 512;      // workgroups of cauchy_image_kernel (256 rows each per sweep of the row space)

}  // namespace

// ------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------
struct bh_hess {
    int64_t d = 0, n = 0, q = 0, q_eff = 0, ld = 0;
    AsyncUpload* up = nullptr;     // non-NULL while bh_hess_create_async's upload may still be running (hess_ready joins it)
    int64_t d_total = 0;           // rows of J over all ranks (= d on one rank): sizes the launch-ahead batch identically everywhere
    int nchunks = 0;
    double mu = 0.0;
    double* Jd = nullptr;          // (d + q) x ld row-major
    int64_t Jd_doubles = 0;        // capacity of the allocation behind Jd (it may come from the image pool)
    double* vpad = nullptr;        // ld
    double* zpad = nullptr;        // ld
    double* upad = nullptr;        // d + q   (J'u input staging / J v output staging)
    double* tbuf = nullptr;        // d + q   (t = J v between the two passes of a column-panel H*p; NULL for n <= 16384)
    double* timg = nullptr;        // 2 x (d + q) + 2 x kCauchyImgGrid: t_d = J~ d, t_s = J~ s_c and the partial sums of the image-space Cauchy search (lazy)
    double* timg_gen = nullptr;    // (1 + mA) x (d + q): a = J~ D g and B = J~ D A' of its linear-equality form (lazy, grown on demand)
    int64_t timg_gen_doubles = 0;
    double* partials = nullptr;    // g_cap x ld
    double* sq_partials = nullptr; // 2 x g_cap (second half: per-workgroup minima of the two-kernel CG iteration)
    double* scalar = nullptr;      // 2
    int g_cap = 0;
    int last_n_hmul = 0;           // H*p count of the previous bh_pcg on this handle (launch schedule hint)
    // tie log of the previous projected_cg on this handle (bh_pcg_tie_info)
    int tie_flags = 0, tie_first = 0, margin_kind = 0, margin_at = 0;
    double min_margin = 0.0;
    unsigned tie_tag = 0;          // tag of the call the log belongs to
    bool tie_read = true;          // false: the words have not been fetched from the host-mapped page yet
    bh_stats_t stats{};
    std::vector<hipEvent_t> ev;    // 2*kEvCap, created lazily
    std::vector<int> ev_pending;   // launch index (within the running bh_pcg) of each recorded pair
    uint64_t hmul_seq = 0;         // H*p launches of bh_pcg calls on this handle (profile sampling)
    bool counted = false;          // included in g_ctx.live_hess (false while a create is failing)
    bool pending_finish = false;   // bh_hess_create_async on a rank without rows: the d_total collective runs at wait / first use, as on its peers
};

struct bh_proj {
    int64_t mA = 0, n = 0, ldA = 0;
    double* Ad = nullptr;          // mA x ldA row-major
    int nfix = 0, mpp = 0;
    bool active_set = false;
    int* fixrank = nullptr;        // ldA ints
    int* fixidx = nullptr;         // n ints
    double* L = nullptr;           // mpp x mpp: the caller's augmented factor (reference form)
    int64_t L_cap = 0;
    bool have_L = false;
    double* M = nullptr;           // mA x mA: A_free A_free' (lower triangle), kept for rank-one downdates (bh_cauchy_step)
    double* Lr = nullptr;          // mA x mA + mA: chol(M) and its reciprocal diagonal (reduced form)
    int* info = nullptr;           // device flag of the Cholesky kernels (= counts + 4)
    double* tpart = nullptr;       // (ldA/32 + 1) x mA: per-workgroup partials of A_free r (three- / four-kernel CG iteration)
    double* W = nullptr;           // 2 x 64 x 64: [Linv | Linv'] of the reduced factor, mA <= 64 (tri_inv_small_kernel)
    bool linv_valid = false;       // W belongs to the current Lr
    int last_cauchy_passes = 0;    // passes of the previous bh_cauchy_step on this handle (decides whether 1 + mA set-up sweeps pay off)
    bool reduced = false;          // form used by bh_project / bh_pcg for the current active set
    bool M_valid = false;          // M = A_free A_free' for the CURRENT active set (false after factor-only downdates)
    std::vector<uint64_t> last_chunks;   // fixvars of the last successful bh_proj_set_active (reduced form: skip identical pushes)
    int* newidx = nullptr;         // n ints: variables fixed by the last bh_proj_update_active_dev, index order
    int* counts = nullptr;         // 8 ints (AU_*)
    unsigned long long* chunks_dev = nullptr;   // ceil(n/64) + 1 words: BitVector image of the device-side mask
    double* tw = nullptr;          // n + 16
    double* rpad = nullptr;        // ldA
    double* vtmp = nullptr;        // ldA
};

namespace {

int32_t ensure_cg_workspace(int64_t n_pad, int64_t trace_cap) {
    CgWorkspace& c = g_ctx.cg;
    if (c.n_pad < n_pad) {
        // order matters: the host-pointer entry points stage (g, wl, wu), (s, x, xlow, xupp, g) or (x, xlow, xupp, g) with ONE DMA
        double** vecs[] = {&c.w, &c.r, &c.v, &c.p, &c.Hp, &c.s, &c.x, &c.xlow, &c.xupp, &c.g, &c.wl, &c.wu, &c.hw, &c.p2, &c.gpart, &c.rvpart,
                           &c.r2, &c.hpx, &c.hpx_tail};
        constexpr int NV = 19;
        dev_free(c.slab);
        c.slab = nullptr;
        g_ctx.hw_note = {};                      // cg.hw goes with the old slab
        BH_TRY(dev_alloc(&c.slab, NV * n_pad + 8));
        BH_HIP(hipMemsetAsync(c.slab, 0, (size_t)(NV * n_pad + 8) * sizeof(double), g_ctx.stream));
        for (int i = 0; i < NV; ++i) *vecs[i] = c.slab + (int64_t)i * n_pad;
        c.scalars = c.slab + NV * n_pad;
        c.n_pad = n_pad;
    }
    if (!c.d_state) {
        BH_TRY(dev_alloc(&c.d_state, 1));
        BH_HIP(hipMemsetAsync(c.d_state, 0, sizeof(CgState), g_ctx.stream));
        void* hp = nullptr;
        BH_HIP(hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent));
        memset(hp, 0, 64);
        c.h_mirror = reinterpret_cast<volatile unsigned long long*>(hp);
        void* dp = nullptr;
        BH_HIP(hipHostGetDevicePointer(&dp, hp, 0));
        c.d_mirror = reinterpret_cast<unsigned long long*>(dp);
    }
    if (trace_cap > c.trace_cap) {
        dev_free(c.d_trace);
        c.d_trace = nullptr;
        BH_TRY(dev_alloc(&c.d_trace, 4 * trace_cap));
        c.trace_cap = trace_cap;
    }
    return BH_OK;
}

struct MirrorWord { int done, status, iter, n_hmul; };
int32_t check_peer_error();
int32_t hess_ready(bh_hess* H);

// Spin on the host-mapped progress word until this call's tag shows `done` or at least `target` H*p products.
// iter_target > 0: also return once the reference's `iter` has reached it (two-kernel box iteration: the H*p launch of
// iteration j publishes iter = j when its prologue has decided that the loop goes on).
int32_t wait_mirror(CgWorkspace& c, unsigned tag, int target, MirrorWord* out, int iter_target = 0) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long spins = 0;
    auto reached = [&](unsigned long long w, MirrorWord* m) {
        if (((w >> 48) & 0xffffu) != (tag & 0xffffu)) return false;
        m->status = (int)((w >> 44) & 0xf);
        m->done = (int)((w >> 40) & 0xf);
        m->iter = (int)((w >> 20) & 0xfffff);
        m->n_hmul = (int)(w & 0xfffff);
        return m->done != 0 || m->n_hmul >= target || (iter_target > 0 && m->iter >= iter_target);
    };
    while (true) {
        if (reached(*c.h_mirror, out)) return BH_OK;
        __builtin_ia32_pause();
        if ((++spins & 0xffff) == 0) {
            if (g_ctx.peer.active && *g_ctx.peer.h_err != 0ull) return check_peer_error();
            hipError_t q = hipStreamQuery(g_ctx.stream);
            if (q != hipSuccess && q != hipErrorNotReady) return fail(BH_ERR_HIP, std::string("CG loop: ") + hipGetErrorString(q));
            if (q == hipSuccess) {
                // everything enqueued has run: the word is final.  Not reaching the target now is a logic error, not a wait.
                MirrorWord m2{};
                if (!reached(*c.h_mirror, &m2)) return fail(BH_ERR_HIP, "internal: stream drained but the loop state did not reach the launch target");
                continue;
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                return fail(BH_ERR_HIP, "CG loop: no progress for 120 s", /*drain=*/false);
        }
    }
}

bool comm_active() { return g_ctx.comm != nullptr || g_ctx.peer.active; }
// the peer-buffer exchange is used when it is the only communicator, or when option "comm_path" = 1 selects it
bool use_peer_path() { return g_ctx.peer.active && (g_ctx.comm_path == 1 || g_ctx.comm == nullptr); }

// buf[0, count) <- sum over ranks, in place, on the library stream.  `buf` must be addressable in whole 16-byte chunks
// (every caller passes a padded workspace vector).  state: the device-side gate of a CG / Cauchy loop — honoured by the
// peer-buffer path; an RCCL call cannot be gated from the device, so over-launched iterations still pay for it there.
int32_t allreduce_inplace(double* buf, int64_t count, bh_hess* H, const CgState* state = nullptr) {
    if (!comm_active() || count <= 0) return BH_OK;
    if (use_peer_path()) {
        for (int64_t off = 0; off < count; off += kPeerCap) {
            const int nch = (int)((std::min(kPeerCap, count - off) + 1) / 2);
            hipLaunchKernelGGL(reduce_exchange_kernel, dim3((nch + kPeerBlockChunks - 1) / kPeerBlockChunks), dim3(256), 0, g_ctx.stream,
                               (const double*)nullptr, (int64_t)0, nch, 0, buf + off, state, g_ctx.peer.args);
        }
        BH_HIP(hipGetLastError());
    } else {
        BH_NCCL(g_ctx.p_ncclAllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, g_ctx.comm, g_ctx.stream));
    }
    if (H) H->stats.n_allreduce += 1;
    return BH_OK;
}

int32_t check_peer_error() {
    if (g_ctx.peer.active && g_ctx.peer.h_err != nullptr && *g_ctx.peer.h_err != 0ull)
        return fail(BH_ERR_RCCL, "peer-buffer all-reduce: a rank did not arrive within the timeout (exchange #" +
                                     std::to_string(*g_ctx.peer.h_err) + "); results are invalid", /*drain=*/false);
    return BH_OK;
}

// z_out (ld doubles) = sum over ranks of the fixed-order sum of the `grid` slab rows a row-stream launch left in
// H->partials.  One launch either way on one rank; with the peer-buffer communicator the exchange is fused into the slab
// reduction (reduce_exchange_kernel), with RCCL it is a separate collective behind reduce_partials_kernel.
int32_t reduce_slabs(bh_hess* H, int grid, double* z_out, const CgState* state) {
    const int nblk = (H->nchunks + kPeerBlockChunks - 1) / kPeerBlockChunks;
    if (use_peer_path() && nblk <= kPeerBlkCap) {
        hipLaunchKernelGGL(reduce_exchange_kernel, dim3(nblk), dim3(256), 0, g_ctx.stream, (const double*)H->partials, H->ld, H->nchunks,
                           grid, z_out, state, g_ctx.peer.args);
        BH_HIP(hipGetLastError());
        H->stats.n_allreduce += 1;
        return BH_OK;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(nblk), dim3(256), 0, g_ctx.stream, H->partials, H->ld, H->nchunks, grid, z_out, state);
    BH_HIP(hipGetLastError());
    return allreduce_inplace(z_out, H->n, H, state);
}

// Column panels: a row wider than the register-resident kernels can hold (ld/2 > kMaxChunks) is swept in panels of
// kPanelChunks 16-byte chunks, one launch per panel; H*p then costs two passes over J (J v accumulated over panels into
// tbuf, then J' (W .* t) per panel) instead of the fused single pass.
constexpr int kPanelChunks = 2048;
constexpr int kPanelCfg = 4;          // <256,8,4>: handles any panel up to 2048 chunks, same grid for every panel

bool multi_panel(const bh_hess* H) { return H->nchunks > kMaxChunks; }

RowStreamArgs rs_args(bh_hess* H, int64_t nrows, const CgState* state) {
    RowStreamArgs a{};
    a.J = H->Jd; a.ld = H->ld; a.nrows = nrows; a.d_rows = H->d; a.nchunks = H->nchunks;
    a.mu = H->mu; a.state = state;
    return a;
}

// t (nrows) = J[:, all panels] v
int32_t launch_jv_panels(bh_hess* H, const double* v_pad, double* t, int64_t nrows, const CgState* state) {
    const int grid = grid_for(kPanelCfg, nrows);
    for (int c0 = 0, p = 0; c0 < H->nchunks; c0 += kPanelChunks, ++p) {
        RowStreamArgs a = rs_args(H, nrows, state);
        a.J = H->Jd + 2 * (int64_t)c0; a.v = v_pad + 2 * (int64_t)c0;
        a.nchunks = std::min(kPanelChunks, H->nchunks - c0);
        a.t_out = t; a.accumulate = p > 0 ? 1 : 0;
        launch_row_stream(kPanelCfg, MODE_JV, a, grid, g_ctx.stream);
    }
    BH_HIP(hipGetLastError());
    return BH_OK;
}

// z_out (ld) = J[:, all panels]' (u or W.*u), reduced over workgroups
int32_t launch_jtv_panels(bh_hess* H, const double* u, double* z_out, int64_t nrows, bool weighted, const CgState* state) {
    const int grid = grid_for(kPanelCfg, nrows);
    for (int c0 = 0; c0 < H->nchunks; c0 += kPanelChunks) {
        RowStreamArgs a = rs_args(H, nrows, state);
        a.J = H->Jd + 2 * (int64_t)c0; a.partials = H->partials + 2 * (int64_t)c0;
        a.nchunks = std::min(kPanelChunks, H->nchunks - c0);
        a.u = u; a.weighted_u = weighted ? 1 : 0;
        launch_row_stream(kPanelCfg, MODE_JTV, a, grid, g_ctx.stream);
    }
    BH_HIP(hipGetLastError());
    return reduce_slabs(H, grid, z_out, state);       // includes the all-reduce over ranks
}

// BH_FLAG_PROFILE: records the start event of an H*p launch when this launch is sampled; *slot >= 0 then names the pair.
int32_t profile_begin(bh_hess* H, int ev_index, int* slot) {
    *slot = -1;
    if (!(g_ctx.flags & BH_FLAG_PROFILE) || ev_index < 0) return BH_OK;
    if ((H->hmul_seq++ % (uint64_t)g_ctx.opt_ev_stride) != 0 || (int)H->ev_pending.size() >= kEvCap) return BH_OK;
    if (H->ev.empty()) {
        H->ev.resize(2 * kEvCap, nullptr);
        for (auto& e : H->ev) BH_HIP(hipEventCreate(&e));
    }
    *slot = (int)H->ev_pending.size();
    H->ev_pending.push_back(ev_index);
    BH_HIP(hipEventRecord(H->ev[2 * *slot], g_ctx.stream));
    return BH_OK;
}

// z_out (ld doubles, device) = sum over ranks of J_k'(W .* (J_k v)), v = v_pad (ld doubles, zero padded).
int32_t launch_hmul(bh_hess* H, const double* v_pad, double* z_out, const CgState* state, int ev_index, int reverse = 0,
                    bool negate = false, const int* negmask = nullptr) {
    BH_TRY(hess_ready(H));
    const int64_t nrows = H->d + H->q_eff;
    if (multi_panel(H)) {
        BH_TRY(launch_jv_panels(H, v_pad, H->tbuf, nrows, state));
        BH_TRY(launch_jtv_panels(H, H->tbuf, z_out, nrows, true, state));
        return BH_OK;
    }
    const int cfg = pick_config(H->nchunks);
    const int grid = grid_for(cfg, nrows);
    RowStreamArgs a = rs_args(H, nrows, state);
    a.v = v_pad; a.partials = H->partials; a.reverse = reverse;
    a.negate = negate ? 1 : 0; a.negmask = negmask;
    // BH_FLAG_PROFILE: hipEvents around every opt_ev_stride-th (default 8th) H*p launch of this handle, counted ACROSS calls
    // (an event pair costs ~10 us of stream time; timing every launch would slow the loop it measures by 3 %).
    int slot = -1;
    BH_TRY(profile_begin(H, ev_index, &slot));
    launch_row_stream(cfg, MODE_FUSED, a, grid, g_ctx.stream);
    if (slot >= 0) BH_HIP(hipEventRecord(H->ev[2 * slot + 1], g_ctx.stream));
    return reduce_slabs(H, grid, z_out, state);
}

int32_t launch_jv(bh_hess* H, const double* v_pad, double* t_out, bool with_c_rows, double* sq_out_scalar) {
    BH_TRY(hess_ready(H));
    const int64_t nrows = H->d + (with_c_rows ? H->q_eff : 0);
    if (multi_panel(H)) {
        double* t = t_out ? t_out : H->tbuf;
        BH_TRY(launch_jv_panels(H, v_pad, t, nrows, nullptr));
        if (sq_out_scalar)
            hipLaunchKernelGGL(weighted_sqsum_kernel, dim3(1), dim3(1024), 0, g_ctx.stream, (const double*)t, nrows, H->d, H->mu, sq_out_scalar);
        BH_HIP(hipGetLastError());
        return BH_OK;
    }
    const int cfg = pick_config(H->nchunks);
    const int grid = grid_for(cfg, nrows);
    RowStreamArgs a = rs_args(H, nrows, nullptr);
    a.v = v_pad; a.t_out = t_out;
    a.sq_partials = sq_out_scalar ? H->sq_partials : nullptr;
    launch_row_stream(cfg, MODE_JV, a, grid, g_ctx.stream);
    if (sq_out_scalar) {
        hipLaunchKernelGGL(reduce_scalar_kernel, dim3(1), dim3(256), 0, g_ctx.stream, H->sq_partials, grid, sq_out_scalar);
    }
    BH_HIP(hipGetLastError());
    return BH_OK;
}

int32_t launch_jtv(bh_hess* H, const double* u_dev, double* z_out, bool with_c_rows = false) {
    BH_TRY(hess_ready(H));
    const int64_t nrows = H->d + (with_c_rows ? H->q_eff : 0);
    if (multi_panel(H)) {
        BH_TRY(launch_jtv_panels(H, u_dev, z_out, nrows, false, nullptr));
        return BH_OK;
    }
    const int cfg = pick_config(H->nchunks);
    const int grid = grid_for(cfg, nrows);
    RowStreamArgs a = rs_args(H, nrows, nullptr);
    a.u = u_dev; a.partials = H->partials;
    launch_row_stream(cfg, MODE_JTV, a, grid, g_ctx.stream);
    return reduce_slabs(H, grid, z_out, nullptr);
}

int32_t alloc_hess_common(bh_hess* H) {
    const int64_t rows = H->d + H->q;
    H->ld = round_up(std::max<int64_t>(H->n, 1), 16);
    H->nchunks = (int)(H->ld / 2);
    H->q_eff = (g_ctx.rank == 0) ? H->q : 0;   // C is replicated: only rank 0 contributes C'(mu C v)
    {   // the image: from the pool when a retired one fits (>= the size needed, <= twice)
        const int64_t need = std::max<int64_t>(rows, 1) * H->ld;
        int best = -1;
        for (int i = 0; i < (int)g_ctx.image_pool.size(); ++i) {
            const int64_t have = g_ctx.image_pool[(size_t)i].doubles;
            if (have >= need && have <= 2 * need && (best < 0 || have < g_ctx.image_pool[(size_t)best].doubles)) best = i;
        }
        if (best >= 0) {
            H->Jd = g_ctx.image_pool[(size_t)best].ptr;
            H->Jd_doubles = g_ctx.image_pool[(size_t)best].doubles;
            g_ctx.image_pool.erase(g_ctx.image_pool.begin() + best);
            g_ctx.image_pool_hits += 1;
        } else {
            BH_TRY(dev_alloc(&H->Jd, need));
            H->Jd_doubles = need;
        }
    }
    BH_TRY(dev_alloc(&H->vpad, H->ld));
    BH_TRY(dev_alloc(&H->zpad, H->ld));
    BH_TRY(dev_alloc(&H->upad, std::max<int64_t>(rows, 1)));
    if (multi_panel(H)) BH_TRY(dev_alloc(&H->tbuf, std::max<int64_t>(rows, 1)));
    BH_HIP(hipMemsetAsync(H->vpad, 0, H->ld * sizeof(double), g_ctx.stream));
    BH_HIP(hipMemsetAsync(H->zpad, 0, H->ld * sizeof(double), g_ctx.stream));
    // partial slabs: enough for the largest grid any variant may use
    int64_t gmax = (int64_t)g_ctx.n_cu * kMaxBlocksPerCu;
    H->g_cap = (int)gmax;
    BH_TRY(dev_alloc(&H->partials, gmax * H->ld));
    // [0, gmax): sum w (Jp)^2 per workgroup; [gmax, 2 gmax): its factor_to_boundary minimum; [2 gmax, 3 gmax): the same, ping-pong
    // partner for the RCCL form of the two-kernel iteration (there the H*p launch reads the previous launch's minima itself)
    BH_TRY(dev_alloc(&H->sq_partials, 3 * gmax));
    BH_TRY(dev_alloc(&H->scalar, 2));
    H->stats.bytes_per_hmul = (multi_panel(H) ? 16.0 : 8.0) * (double)(H->d + H->q_eff) * (double)H->n + 16.0 * (double)H->n;
    return BH_OK;
}

int32_t finish_hess_create(bh_hess* H);

static void async_upload_destroy(AsyncUpload* u) {
    if (!u) return;
    for (int i = 0; i < 2; ++i) {
        if (u->copied[i]) (void)hipEventDestroy(u->copied[i]);
        if (u->freed[i]) (void)hipEventDestroy(u->freed[i]);
        dev_free(u->staging[i]);
    }
    if (u->s_copy) (void)hipStreamDestroy(u->s_copy);
    if (u->s_xpose) (void)hipStreamDestroy(u->s_xpose);
    delete u;
}
// A finished upload hands its streams / events / staging buffers to the next one (creating them costs ~3 ms, hipFree of the
// staging buffers another ~3 ms: more than a 64 MiB upload itself).
static void async_upload_cleanup(AsyncUpload* u) {
    if (g_ctx.init && g_ctx.upload_cache == nullptr && u->resources() && u->rc == BH_OK) {
        u->detail.clear();
        g_ctx.upload_cache = u;
        return;
    }
    async_upload_destroy(u);
}

// Worker of bh_hess_create_async.  Never touches g_ctx.detail (the caller's thread owns it): errors travel in u.rc / u.detail.
static void async_upload_worker(bh_hess* H, const double* J, int64_t d, int64_t n, int64_t ldJ, int device) {
    AsyncUpload& u = *H->up;
    auto bad = [&](const char* what, hipError_t e) { u.rc = BH_ERR_HIP; u.detail = std::string(what) + ": " + hipGetErrorString(e); };
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { bad("hipSetDevice (upload thread)", e); return; }
    const int64_t cc = u.chunk_cols;
    int k = 0;
    for (int64_t c0 = 0; c0 < n; c0 += cc, ++k) {
        const int slot = k & 1;
        const int64_t cols = std::min(cc, n - c0);
        const bool last = c0 + cols >= n;
        // the staging slot is free once the transpose that read it (two chunks ago) has finished
        if (k >= 2 && (e = hipEventSynchronize(u.freed[slot])) != hipSuccess) { bad("hipEventSynchronize", e); return; }
        if (ldJ == d) e = hipMemcpyAsync(u.staging[slot], J + c0 * ldJ, (size_t)d * cols * sizeof(double), hipMemcpyHostToDevice, u.s_copy);
        else e = hipMemcpy2DAsync(u.staging[slot], (size_t)d * sizeof(double), J + c0 * ldJ, (size_t)ldJ * sizeof(double),
                                  (size_t)d * sizeof(double), (size_t)cols, hipMemcpyHostToDevice, u.s_copy);
        if (e != hipSuccess) { bad("hipMemcpyAsync (J chunk)", e); return; }
        if ((e = hipEventRecord(u.copied[slot], u.s_copy)) != hipSuccess) { bad("hipEventRecord", e); return; }
        if ((e = hipStreamWaitEvent(u.s_xpose, u.copied[slot], 0)) != hipSuccess) { bad("hipStreamWaitEvent", e); return; }
        const int64_t wcols = last ? H->ld - c0 : cols;          // the last chunk also writes the zero padding of every row
        dim3 grid((unsigned)((d + 31) / 32), (unsigned)((wcols + 31) / 32));
        hipLaunchKernelGGL(transpose_cm_to_rm_kernel, grid, dim3(256), 0, u.s_xpose, (const double*)u.staging[slot], d, d, cols,
                           H->Jd + c0, H->ld, wcols);
        if ((e = hipGetLastError()) != hipSuccess) { bad("transpose launch", e); return; }
        if ((e = hipEventRecord(u.freed[slot], u.s_xpose)) != hipSuccess) { bad("hipEventRecord", e); return; }
    }
    if ((e = hipStreamSynchronize(u.s_xpose)) != hipSuccess) bad("hipStreamSynchronize (upload)", e);
}

// Every entry point that reads the image calls this first: joins a pending asynchronous upload (bh_hess_wait does the same).
int32_t hess_ready(bh_hess* H) {
    if (H && !H->up && H->pending_finish) { H->pending_finish = false; return finish_hess_create(H); }
    if (!H || !H->up) return BH_OK;
    AsyncUpload* u = H->up;
    if (u->worker.joinable()) u->worker.join();
    const int32_t rc = u->rc;
    const std::string detail = u->detail;
    async_upload_cleanup(u);
    H->up = nullptr;
    if (rc != BH_OK) return fail(rc, "asynchronous J upload: " + detail);
    return finish_hess_create(H);
}

// Last step of every bh_hess constructor.  With a communicator the ranks' row counts are summed once: everything that
// shapes the launch schedule (launch_batch_size) must be computed from data that is identical on all ranks, or ranks would
// enqueue different numbers of iterations — and of collectives (q_eff and the +-1 row of row_shard differ between ranks).
int32_t finish_hess_create(bh_hess* H) {
    H->d_total = H->d;
    if (comm_active()) {
        double v[2] = {(double)H->d, 0.0};
        BH_HIP(hipMemcpyAsync(H->scalar, v, sizeof(v), hipMemcpyHostToDevice, g_ctx.stream));
        BH_TRY(allreduce_inplace(H->scalar, 1, nullptr));
        BH_HIP(hipMemcpyAsync(v, H->scalar, sizeof(double), hipMemcpyDeviceToHost, g_ctx.stream));
        BH_HIP(hipStreamSynchronize(g_ctx.stream));
        BH_TRY(check_peer_error());
        H->d_total = (int64_t)std::llround(v[0]);
    }
    return BH_OK;
}

// Upload a column-major host matrix (rows x cols, leading dimension ldh) into rows [row0, row0+rows) of a
// row-major padded device image with leading dimension ldd.
// The same from a column-major matrix that already lives in HBM (a device-side jac_res): transpose only, no PCIe.
int32_t transpose_from_device(const double* src_dev, int64_t rows, int64_t cols, int64_t lds, double* dst_image, int64_t row0, int64_t ldd) {
    if (rows == 0) return BH_OK;
    dim3 grid((unsigned)((rows + 31) / 32), (unsigned)((ldd + 31) / 32));
    hipLaunchKernelGGL(transpose_cm_to_rm_kernel, grid, dim3(256), 0, g_ctx.stream, src_dev, lds, rows, cols, dst_image + row0 * ldd, ldd, ldd);
    BH_HIP(hipGetLastError());
    return BH_OK;
}

int32_t upload_transposed(const double* host, int64_t rows, int64_t cols, int64_t ldh, double* dst_image, int64_t row0,
                          int64_t ldd) {
    if (rows == 0) return BH_OK;
    double* staging = nullptr;
    BH_TRY(dev_alloc(&staging, rows * std::max<int64_t>(cols, 1)));
    if (cols > 0) {
        count_h2d((size_t)rows * cols * sizeof(double));
        hipError_t e = hipMemcpy2DAsync(staging, (size_t)rows * sizeof(double), host, (size_t)ldh * sizeof(double),
                                        (size_t)rows * sizeof(double), (size_t)cols, hipMemcpyHostToDevice, g_ctx.stream);
        if (e != hipSuccess) { dev_free(staging); return fail(BH_ERR_HIP, std::string("hipMemcpy2DAsync: ") + hipGetErrorString(e)); }
    }
    dim3 grid((unsigned)((rows + 31) / 32), (unsigned)((ldd + 31) / 32));
    hipLaunchKernelGGL(transpose_cm_to_rm_kernel, grid, dim3(256), 0, g_ctx.stream, staging, rows, rows, cols,
                       dst_image + row0 * ldd, ldd, ldd);
    hipError_t e = hipStreamSynchronize(g_ctx.stream);
    dev_free(staging);
    if (e != hipSuccess) return fail(BH_ERR_HIP, std::string("transpose: ") + hipGetErrorString(e));
    return BH_OK;
}

// Host <-> device vector traffic goes through a pinned arena: the caller's arrays are pageable (Julia / NumPy heap), and a
// pageable hipMemcpyAsync costs 20-25 us per 32 KiB vector; memcpy into pinned memory + a truly asynchronous DMA costs ~4.
// The arena is a bump allocator reset by sync_flush(), which every export calls before returning (host pointers are only
// borrowed for the duration of the call); results are copied out of the arena after the stream has drained.
struct PendingOut { double* dst; const double* pin; int64_t n; };
struct PinArena {
    char* base = nullptr;
    char* dev_base = nullptr;      // the same pages as the GPU sees them (pinned host memory is mapped into the device's address space)
    size_t cap = 0, used = 0;
    std::vector<PendingOut> outs;
    bool dma = false;              // a DMA engine transfer has been enqueued since the last flush (then the flush must be a real drain)
};
PinArena g_pin;
inline void note_dma() { g_pin.dma = true; }
// Small host <-> device vector copies done by a KERNEL on the mapped arena instead of a DMA engine: a 32 KiB hipMemcpyAsync costs
// ~10 us of stream time and a kernel behind it another cross-engine dependency; a copy kernel reads / writes the pinned pages
// over PCIe in ~3 us and keeps everything on the compute queue, so that the call can end with a mailbox seal + poll instead
// of a hipStreamSynchronize.
__global__ __launch_bounds__(256) void arena_copy_kernel(double* __restrict__ dst, const double* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}
inline double* pin_dev_view(const double* pin) { return reinterpret_cast<double*>(g_pin.dev_base + (reinterpret_cast<const char*>(pin) - g_pin.base)); }
inline bool arena_kernels() { return g_ctx.opt_host_copy_kernels != 0 && g_pin.dev_base != nullptr; }
inline void launch_arena_copy(double* dst, const double* src, int64_t n) {
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256));
    hipLaunchKernelGGL(arena_copy_kernel, dim3(grid), dim3(256), 0, g_ctx.stream, dst, src, n);
}
bool pin_arena_busy() { return g_pin.used > 0; }
void pin_arena_abandon(bool drained) {
    g_pin.outs.clear();
    g_pin.used = 0;
    // transfers may still be in flight (the stream could not be drained): retire this arena — it is leaked on purpose,
    // a later pin_alloc maps a fresh one — rather than let the next call's vectors share memory with a stale DMA
    if (!drained) { g_pin.base = nullptr; g_pin.dev_base = nullptr; g_pin.cap = 0; }
    g_pin.dma = false;
}
constexpr size_t kPinArenaBytes = 32u << 20;
constexpr int64_t kPinMaxVec = 1 << 20;     // doubles; larger transfers go directly (bandwidth-, not latency-bound)

double* pin_alloc(int64_t n) {
    if (n > kPinMaxVec) return nullptr;
    if (!g_pin.base) {
        void* p = nullptr;
        if (hipHostMalloc(&p, kPinArenaBytes, hipHostMallocDefault) != hipSuccess) return nullptr;
        g_pin.base = static_cast<char*>(p);
        g_pin.cap = kPinArenaBytes;
        void* dp = nullptr;
        g_pin.dev_base = (hipHostGetDevicePointer(&dp, p, 0) == hipSuccess) ? static_cast<char*>(dp) : nullptr;
        if (g_pin.dev_base == nullptr) (void)hipGetLastError();
    }
    const size_t bytes = ((size_t)n * sizeof(double) + 255) & ~(size_t)255;
    if (g_pin.used + bytes > g_pin.cap) return nullptr;
    double* r = reinterpret_cast<double*>(g_pin.base + g_pin.used);
    g_pin.used += bytes;
    return r;
}

int32_t stage_vec(double* dst_pad, const double* src, int64_t n, bool src_is_device) {
    if (n == 0) return BH_OK;
    if (!src_is_device) count_h2d((size_t)n * sizeof(double));
    if (!src_is_device) {
        if (double* pin = pin_alloc(n)) {
            memcpy(pin, src, (size_t)n * sizeof(double));
            if (arena_kernels()) { launch_arena_copy(dst_pad, pin_dev_view(pin), n); BH_HIP(hipGetLastError()); return BH_OK; }
            note_dma();
            BH_HIP(hipMemcpyAsync(dst_pad, pin, (size_t)n * sizeof(double), hipMemcpyHostToDevice, g_ctx.stream));
            return BH_OK;
        }
        note_dma();
    }
    BH_HIP(hipMemcpyAsync(dst_pad, src, (size_t)n * sizeof(double), src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                          g_ctx.stream));
    return BH_OK;
}

// A device vector the kernels may read in whole 16-byte chunks up to the padded length: the caller's own buffer when it needs no
// padding (n == n_pad) and is 16-byte aligned, else a zero-padded staging copy (pad must be zero beyond n already).
int32_t device_operand(const double** out, double* pad, const double* v_dev, int64_t n, int64_t n_pad) {
    if (n == n_pad && (reinterpret_cast<uintptr_t>(v_dev) & 15u) == 0) { *out = v_dev; return BH_OK; }
    BH_TRY(stage_vec(pad, v_dev, n, true));
    *out = pad;
    return BH_OK;
}

// Several host vectors into CONSECUTIVE workspace vectors (stride n_pad) with one DMA instead of one each: a 32 KiB
// hipMemcpyAsync costs ~10 us of stream time whatever its size, and the host-pointer entry points stage 3-5 of them.
int32_t stage_vecs(double* dst_first, std::initializer_list<const double*> srcs, int64_t n, int64_t n_pad) {
    const int64_t k = (int64_t)srcs.size();
    double* pin = (n > 0) ? pin_alloc(k * n_pad) : nullptr;
    if (pin == nullptr) {
        int64_t i = 0;
        for (const double* src : srcs) {       // too large for the arena: one copy each; the padding is zeroed as the batched path does
            BH_TRY(stage_vec(dst_first + i * n_pad, src, n, false));
            if (n_pad > n) BH_HIP(hipMemsetAsync(dst_first + i * n_pad + n, 0, (size_t)(n_pad - n) * sizeof(double), g_ctx.stream));
            ++i;
        }
        return BH_OK;
    }
    int64_t i = 0;
    for (const double* src : srcs) {
        memcpy(pin + i * n_pad, src, (size_t)n * sizeof(double));
        if (n_pad > n) memset(pin + i * n_pad + n, 0, (size_t)(n_pad - n) * sizeof(double));
        ++i;
    }
    count_h2d((size_t)k * n * sizeof(double));
    if (arena_kernels()) { launch_arena_copy(dst_first, pin_dev_view(pin), k * n_pad); BH_HIP(hipGetLastError()); return BH_OK; }
    note_dma();
    BH_HIP(hipMemcpyAsync(dst_first, pin, (size_t)(k * n_pad) * sizeof(double), hipMemcpyHostToDevice, g_ctx.stream));
    return BH_OK;
}

int32_t fetch_vec(double* dst, const double* src_dev, int64_t n, bool dst_is_device) {
    if (n == 0) return BH_OK;
    if (!dst_is_device) count_d2h((size_t)n * sizeof(double));
    if (!dst_is_device) {
        if (double* pin = pin_alloc(n)) {
            if (arena_kernels()) { launch_arena_copy(pin_dev_view(pin), src_dev, n); BH_HIP(hipGetLastError()); }
            else { note_dma(); BH_HIP(hipMemcpyAsync(pin, src_dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, g_ctx.stream)); }
            g_pin.outs.push_back({dst, pin, n});
            return BH_OK;
        }
        note_dma();
    }
    BH_HIP(hipMemcpyAsync(dst, src_dev, (size_t)n * sizeof(double), dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                          g_ctx.stream));
    return BH_OK;
}

// ---- mailbox ------------------------------------------------------------------------------
constexpr size_t kMboxBytes = 64 * 1024;
constexpr size_t kMboxPayloadOff = 64;
constexpr size_t kMbScal = 0, kMbInts = 16, kMbChunks = 64;          // payload layout: 2 doubles | 8 ints | BitVector image
// Optional riders: up to two doubles and `icount` ints that live in device memory (results an all-reduce or a later kernel
// also needs there) are copied into the page by the sealing thread itself.
__global__ void mbox_seal_kernel(unsigned long long* seq_word, unsigned long long seq, const double* a, const double* b, double* ddst,
                                 const int* isrc, int icount, int* idst) {
    // runs after the producers (same stream): their stores to the page were performed before they completed
    if (a != nullptr) ddst[0] = a[0];
    if (b != nullptr) ddst[1] = b[0];
    for (int i = 0; i < icount; ++i) idst[i] = isrc[i];
    __threadfence_system();
    __hip_atomic_store(seq_word, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
int32_t mbox_ensure() {
    if (g_ctx.mbox_h) return BH_OK;
    void* hp = nullptr;
    BH_HIP(hipHostMalloc(&hp, kMboxBytes, hipHostMallocMapped | hipHostMallocCoherent));
    memset(hp, 0, kMboxBytes);
    void* dp = nullptr;
    BH_HIP(hipHostGetDevicePointer(&dp, hp, 0));
    g_ctx.mbox_h = static_cast<volatile unsigned long long*>(hp);
    g_ctx.mbox_d = static_cast<unsigned long long*>(dp);
    g_ctx.mbox_seq = 0;
    return BH_OK;
}
// device / host views of the payload (byte offset `off`, a multiple of 8)
template <class T> T* mbox_dev(size_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(g_ctx.mbox_d) + kMboxPayloadOff + off); }
template <class T> const volatile T* mbox_host(size_t off) {
    return reinterpret_cast<const volatile T*>(reinterpret_cast<const volatile char*>(g_ctx.mbox_h) + kMboxPayloadOff + off);
}
// Enqueue the seal; returns its sequence number through *seq_out.
int32_t mbox_seal(unsigned long long* seq_out, const double* a = nullptr, const double* b = nullptr, double* ddst = nullptr,
                  const int* isrc = nullptr, int icount = 0, int* idst = nullptr) {
    const unsigned long long seq = ++g_ctx.mbox_seq;
    hipLaunchKernelGGL(mbox_seal_kernel, dim3(1), dim3(1), 0, g_ctx.stream, g_ctx.mbox_d, seq, a, b, ddst, isrc, icount, idst);
    BH_HIP(hipGetLastError());
    *seq_out = seq;
    return BH_OK;
}
// Spin until the seal `seq` has been observed: the stream executes in order, so EVERYTHING enqueued before the seal — kernels,
// DMAs into and out of the pinned arena — has completed by then.
int32_t mbox_poll(unsigned long long seq) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long spins = 0;
    while (g_ctx.mbox_h[0] != seq) {
        __builtin_ia32_pause();
        if ((++spins & 0xffff) == 0) {
            if (g_ctx.peer.active && *g_ctx.peer.h_err != 0ull) return check_peer_error();
            const hipError_t q = hipStreamQuery(g_ctx.stream);
            if (q != hipSuccess && q != hipErrorNotReady) return fail(BH_ERR_HIP, std::string("mailbox wait: ") + hipGetErrorString(q));
            if (q == hipSuccess && g_ctx.mbox_h[0] != seq) return fail(BH_ERR_HIP, "internal: stream drained but the mailbox was not sealed");
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                return fail(BH_ERR_HIP, "mailbox wait: no progress for 120 s", /*drain=*/false);
        }
    }
    // what the device wrote before the seal (mailbox payload, results parked in the pinned arena) is read after it: keep the
    // compiler from hoisting those plain loads above the poll (the hardware keeps load order on its own)
    std::atomic_thread_fence(std::memory_order_acquire);
    return BH_OK;
}
void pin_arena_deliver() {
    for (const PendingOut& o : g_pin.outs) memcpy(o.dst, o.pin, (size_t)o.n * sizeof(double));
    g_pin.outs.clear();
    g_pin.used = 0;
}

// Wait until everything enqueued has completed, deliver the results parked in the pinned arena, reset the arena:
// hipStreamSynchronize.  (Experiment switch mailbox_flush = 1: a seal in the mailbox + a poll instead.  It pays where no DMA
// precedes it — the device-pointer entry points use it through mbox_seal_and_wait — but behind a D2H DMA the seal kernel
// waits for a cross-engine dependency that costs more than the synchronize it replaces: measured slower, off by default.)
int32_t sync_flush() {
    // No DMA engine involved since the last flush (the host vectors travelled through arena_copy_kernel): a seal + poll ends
    // the call without a hipStreamSynchronize.  (mailbox_flush = 1 forces the seal path also behind DMAs: measured slower.)
    const bool kernels_only = arena_kernels() && !g_pin.dma;
    g_pin.dma = false;
    if (!g_ctx.opt_final_sync && (g_ctx.opt_mbox_flush || kernels_only) && g_ctx.mbox_h != nullptr) {
        unsigned long long seq = 0;
        BH_TRY(mbox_seal(&seq));
        BH_TRY(mbox_poll(seq));
        pin_arena_deliver();
        return check_peer_error();
    }
    hipError_t e = hipStreamSynchronize(g_ctx.stream);
    if (e == hipSuccess) pin_arena_deliver();
    g_pin.outs.clear();
    g_pin.used = 0;
    if (e != hipSuccess) return fail(BH_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    return check_peer_error();
}

// Seal what the kernels enqueued so far have written into the mailbox (plus the riders) and wait for it.  `bytes`: payload
// size, for the transfer counters.  Results parked in the pinned arena by a host-pointer caller are delivered as well.
int32_t mbox_seal_and_wait(size_t bytes, const double* a = nullptr, const double* b = nullptr, double* ddst = nullptr,
                           const int* isrc = nullptr, int icount = 0, int* idst = nullptr) {
    unsigned long long seq = 0;
    BH_TRY(mbox_seal(&seq, a, b, ddst, isrc, icount, idst));
    count_d2h(bytes);
    if (g_ctx.opt_final_sync) {
        const hipError_t e = hipStreamSynchronize(g_ctx.stream);
        if (e != hipSuccess) return fail(BH_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    } else {
        BH_TRY(mbox_poll(seq));
    }
    pin_arena_deliver();
    return check_peer_error();
}
// End of a device-pointer call whose results stay in HBM: nothing is owed to the host, later calls are ordered behind this
// one on the library stream (option final_sync = 1: drain as the host-pointer entry points do).
int32_t finish_device_call() {
    if (g_ctx.opt_final_sync || pin_arena_busy()) return sync_flush();
    BH_HIP(hipGetLastError());
    return check_peer_error();
}

// ---- projection -------------------------------------------------------------------------
int32_t ensure_reduced_buffers(bh_proj* P) {
    const int64_t mA = P->mA;
    if (!P->M) BH_TRY(dev_alloc(&P->M, mA * mA));
    if (!P->Lr) BH_TRY(dev_alloc(&P->Lr, mA * mA + mA));
    if (!P->info) P->info = P->counts + 4;      // (rides to the host with the AU_* counts: one copy by the mailbox's sealing thread)
    if (!P->tpart) BH_TRY(dev_alloc(&P->tpart, (P->ldA / 32 + 1) * std::max<int64_t>(mA, 1)));
    if (!P->W && mA <= 64) BH_TRY(dev_alloc(&P->W, 2 * 64 * 64));
    return BH_OK;
}

// Lr = chol(M) (lower, column-major mA x mA).  mA <= 64: one launch of the register-panel kernel (+ reciprocal diagonal
// after the matrix for trsv_small_kernel).  Larger: blocked right-looking — per 64-column panel potrf on the diagonal
// block, trsm for the rows below it, syrk for the trailing matrix (3 launches per panel; 1.6 ms -> ~0.3 ms at mA = 256).
int32_t launch_chol(bh_proj* P, const CgState* gate) {
    const int mA = (int)P->mA;
    hipStream_t s = g_ctx.stream;
    P->linv_valid = false;
    if (mA <= 64) {
        hipLaunchKernelGGL(chol_small_kernel, dim3(1), dim3(256), 0, s, (const double*)P->M, (int64_t)mA, P->Lr, (int64_t)mA, mA,
                           P->Lr + (int64_t)mA * mA, P->info, 0, gate == nullptr ? 1 : 0, gate);
        BH_HIP(hipGetLastError());
        return BH_OK;
    }
    if (!g_ctx.opt_chol_blocked) {
        hipLaunchKernelGGL(chol_lower_kernel, dim3(1), dim3(CG_T), 0, s, (const double*)P->M, P->Lr, mA, P->info, gate);
        BH_HIP(hipGetLastError());
        return BH_OK;
    }
    double* dinv = P->Lr + (int64_t)mA * mA;      // scratch for the current panel's reciprocal diagonal (mA extra doubles)
    if (!g_ctx.trsm_lds_granted) {                 // 97 KiB of dynamic LDS: above what a kernel may use without asking
        BH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&chol_trsm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kTrsmLdsBytes));
        g_ctx.trsm_lds_granted = true;
    }
    hipLaunchKernelGGL(copy_lower_kernel, dim3(std::min(1024, (mA * mA + 255) / 256)), dim3(256), 0, s, (const double*)P->M, P->Lr, mA,
                       P->info, gate);
    for (int k0 = 0; k0 < mA; k0 += 64) {
        const int nb = std::min(64, mA - k0);
        double* blk = P->Lr + k0 + (int64_t)k0 * mA;
        hipLaunchKernelGGL(chol_small_kernel, dim3(1), dim3(256), 0, s, (const double*)blk, (int64_t)mA, blk, (int64_t)mA, nb, dinv, P->info,
                           k0, 0, gate);
        const int rem = mA - k0 - nb;
        if (rem > 0) {
            hipLaunchKernelGGL(chol_trsm_kernel, dim3((rem + TRSM_T - 1) / TRSM_T), dim3(TRSM_T), kTrsmLdsBytes, s, P->Lr, mA, k0, nb,
                               (const double*)dinv, gate);
            const int nt = (rem + 15) / 16;
            hipLaunchKernelGGL(chol_syrk_kernel, dim3(nt, nt), dim3(256), 0, s, P->Lr, mA, k0, nb, gate);
        }
    }
    BH_HIP(hipGetLastError());
    return BH_OK;
}

// M = A_free A_free' from the device-side mask, then Lr = chol(M).  `gate`: skip when gate->done (in-loop use).
int32_t launch_reduced_factor(bh_proj* P, bool use_mask, const CgState* gate) {
    const int mA = (int)P->mA;
    // Measured (n = 4096): VALU kernel 12 us at mA = 64, 107 us at mA = 256; MFMA kernel ~33 us flat up to mA ~ 350 (one
    // workgroup per tile leaves most CUs idle for few tiles) -> matrix cores from mA > 96; opt_gram_mfma = 2 forces them.
    if ((g_ctx.opt_gram_mfma == 1 && mA > 96) || g_ctx.opt_gram_mfma == 2) {
        const int nt = (mA + 15) / 16;            // 16 x 16 lower tiles on the matrix cores (v_mfma_f64_16x16x4_f64)
        hipLaunchKernelGGL(gram_free_mfma_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(GRAM_T), 0, g_ctx.stream, P->Ad, P->ldA, mA,
                           use_mask ? P->fixrank : (const int*)nullptr, P->M);
    } else {
        const int64_t pairs = (int64_t)mA * (mA + 1) / 2;
        hipLaunchKernelGGL(gram_free_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, g_ctx.stream, P->Ad, P->ldA, mA,
                           use_mask ? P->fixrank : (const int*)nullptr, P->M);
    }
    BH_TRY(launch_chol(P, gate));
    P->M_valid = true;
    return BH_OK;
}

ProjArgs proj_args(bh_proj* P, const CgState* st, bool reduced, bool force_mask = false) {
    ProjArgs a{};
    a.A = P->Ad; a.ldA = P->ldA; a.mA = (int)P->mA; a.n = (int)P->n; a.nfix = P->nfix;
    a.fixrank = (P->nfix > 0 || force_mask) ? P->fixrank : nullptr;
    a.fixidx = P->fixidx; a.tw = P->tw; a.state = st;
    a.reduced = reduced ? 1 : 0;
    a.L = reduced ? P->Lr : P->L;
    a.mpp = reduced ? (int)P->mA : (int)P->mA + P->nfix;
    return a;
}

// One pass of the CG loop body: register-resident kernel when the vectors fit (n <= 8192), generic otherwise.
template <int PHASE>
void launch_cg_step(const CgArgs& a, hipStream_t s) {
    const int nch = (a.n + 1) / 2;
    if (nch <= CG_T) hipLaunchKernelGGL((cg_step_reg_kernel<1, PHASE>), dim3(1), dim3(CG_T), 0, s, a);
    else if (nch <= 2 * CG_T) hipLaunchKernelGGL((cg_step_reg_kernel<2, PHASE>), dim3(1), dim3(CG_T), 0, s, a);
    else if (nch <= 4 * CG_T) hipLaunchKernelGGL((cg_step_reg_kernel<4, PHASE>), dim3(1), dim3(CG_T), 0, s, a);
    else hipLaunchKernelGGL((cg_step_kernel<PHASE>), dim3(1), dim3(CG_T), 0, s, a);
}

// First pass of a box-constrained CG with the initialisation folded in (n <= 8192).
void launch_cg_first_step(const CgArgs& a, hipStream_t s) {
    const int nch = (a.n + 1) / 2;
    if (nch <= CG_T) hipLaunchKernelGGL((cg_step_reg_kernel<1, 0, true>), dim3(1), dim3(CG_T), 0, s, a);
    else if (nch <= 2 * CG_T) hipLaunchKernelGGL((cg_step_reg_kernel<2, 0, true>), dim3(1), dim3(CG_T), 0, s, a);
    else hipLaunchKernelGGL((cg_step_reg_kernel<4, 0, true>), dim3(1), dim3(CG_T), 0, s, a);
}

// LDS of trsv_pair_kernel: x[m2] | tile[64*65] | part[4*m2] (only when the split trailing update fits the 160 KiB of a CU)
constexpr size_t kLdsPerCu = 160 * 1024;
int trsv_split_for(int m) {
    const size_t m2 = (size_t)((m + 1) & ~1);
    return (5 * m2 + 64 * 65) * sizeof(double) <= kLdsPerCu ? 4 : 1;
}
// The dynamic-LDS ceiling of the kernel is a property of the function, shared by every handle: only ever raise it.
int32_t ensure_trsv_lds(size_t lds) {
    if (lds <= g_ctx.trsv_lds_granted) return BH_OK;
    BH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trsv_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    g_ctx.trsv_lds_granted = lds;
    return BH_OK;
}
size_t trsv_lds_bytes(int m) {
    const size_t m2 = (size_t)((m + 1) & ~1);
    return ((trsv_split_for(m) == 4 ? 5 : 1) * m2 + 64 * 65) * sizeof(double);
}

// v_out = P(r_pad): r_pad is a zero-padded ldA-length device vector, v_out has >= n entries.
int32_t launch_project(bh_proj* P, const double* r_pad, double* v_out, const CgState* st, bool device_mask = false) {
    // device_mask: the active set lives in P->fixrank on the device and may be ahead of the host's P->nfix (bh_cauchy_step)
    const int n = (int)P->n;
    if (P->mA == 0) {
        const int grid = std::max(1, std::min((n + 255) / 256, 1024));
        hipLaunchKernelGGL(proj_mask_kernel, dim3(grid), dim3(256), 0, g_ctx.stream, r_pad, v_out,
                           (P->nfix > 0 || device_mask) ? P->fixrank : (const int*)nullptr, n, st);
        BH_HIP(hipGetLastError());
        return BH_OK;
    }
    ProjArgs a = proj_args(P, st, device_mask ? true : P->reduced, device_mask);
    const int grid1 = a.mA + (a.reduced ? 0 : (a.nfix + 255) / 256);
    hipLaunchKernelGGL(proj_left_mul_kernel, dim3(grid1), dim3(256), 0, g_ctx.stream, a, r_pad);
    if (a.reduced && a.mpp <= 64) hipLaunchKernelGGL(trsv_small_kernel, dim3(1), dim3(256), 0, g_ctx.stream, a);
    else hipLaunchKernelGGL(trsv_pair_kernel, dim3(1), dim3(CG_T), trsv_lds_bytes(a.mpp), g_ctx.stream, a, trsv_split_for(a.mpp));
    const int nch = (n + 1) / 2;
    if (a.mA <= 64) hipLaunchKernelGGL((proj_left_mul_tr_kernel<true, 4>), dim3((nch + 63) / 64), dim3(256), 0, g_ctx.stream, a, r_pad, v_out);
    else hipLaunchKernelGGL((proj_left_mul_tr_kernel<true, 16>), dim3((nch + 63) / 64), dim3(1024), 0, g_ctx.stream, a, r_pad, v_out);
    BH_HIP(hipGetLastError());
    return BH_OK;
}

int32_t check_proj_ready(bh_proj* P) {
    if (!P) return fail(BH_ERR_INVALID_ARG, "NULL bh_proj");
    if (P->mA > 0 && !P->active_set) return fail(BH_ERR_PRECONDITION, "bh_proj_set_active has not been called (factor missing)");
    return BH_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// exports: library / device
// ------------------------------------------------------------------------------------------
extern "C" {

const char* bh_strerror(int32_t code) {
    switch (code) {
        case BH_OK: return "ok";
        case BH_ERR_INVALID_ARG: return "invalid argument";
        case BH_ERR_NOT_INIT: return "library not initialised (call bh_init)";
        case BH_ERR_HIP: return "HIP runtime error";
        case BH_ERR_RCCL: return "RCCL error";
        case BH_ERR_PRECONDITION: return "precondition of the reference violated";
        case BH_ERR_SHAPE: return "shape mismatch between handles";
        case BH_ERR_NO_DEVICE: return "no usable GPU device";
        case BH_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown error code";
    }
}

const char* bh_last_error_detail(void) { return g_ctx.detail.c_str(); }

int32_t bh_init(int32_t device, int32_t flags) {
    if (g_ctx.init) {
        if (device != g_ctx.device)
            return fail(BH_ERR_INVALID_ARG, "bh_init: already initialised on another device (bh_shutdown first)");
        g_ctx.flags = flags;
        return BH_OK;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(BH_ERR_NO_DEVICE, "hipGetDeviceCount found no device");
    if (device < 0 || device >= count) return fail(BH_ERR_INVALID_ARG, "device ordinal out of range");
    BH_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    BH_HIP(hipGetDeviceProperties(&prop, device));
    g_ctx.n_cu = prop.multiProcessorCount;
    g_ctx.device = device;
    g_ctx.flags = flags;
    BH_HIP(hipStreamCreateWithFlags(&g_ctx.own_stream, hipStreamNonBlocking));
    g_ctx.stream = g_ctx.own_stream;
    BH_TRY(dev_alloc(&g_ctx.scratch_dev, 1024));
    BH_TRY(mbox_ensure());
    if (const char* s = getenv("BH_MAILBOX_FLUSH")) g_ctx.opt_mbox_flush = atoll(s) ? 1 : 0;
    if (const char* s = getenv("BH_HOST_COPY_KERNELS")) g_ctx.opt_host_copy_kernels = atoll(s) ? 1 : 0;
    if (const char* s = getenv("BH_RS_VARIANT")) g_ctx.opt_variant = atoll(s);
    if (const char* s = getenv("BH_BLOCKS_PER_CU")) g_ctx.opt_blocks_per_cu = std::min<int64_t>(std::max<int64_t>(0, atoll(s)), kMaxBlocksPerCu);
    if (const char* s = getenv("BH_PCG_BATCH")) g_ctx.opt_batch = std::max<int64_t>(0, atoll(s));
    if (const char* s = getenv("BH_PINGPONG")) g_ctx.opt_pingpong = atoll(s) ? 1 : 0;
    if (const char* s = getenv("BH_PROJ_FORM")) g_ctx.opt_proj_form = atoll(s) ? 1 : 0;
    if (const char* s = getenv("BH_CG_FUSED")) g_ctx.opt_cg_fused = std::min<int64_t>(std::max<int64_t>(0, atoll(s)), 2);
    if (const char* s = getenv("BH_FINAL_SYNC")) g_ctx.opt_final_sync = atoll(s) ? 1 : 0;
    g_ctx.init = true;
    return BH_OK;
}

static void peer_release(bool with_barrier);

int32_t bh_shutdown(void) {
    if (!g_ctx.init) return BH_OK;
    (void)hipStreamSynchronize(g_ctx.stream);
    if (g_ctx.comm && g_ctx.p_ncclCommDestroy) { g_ctx.p_ncclCommDestroy(g_ctx.comm); g_ctx.comm = nullptr; }
    if (g_ctx.peer.active) peer_release(true);
    g_ctx.comm_path = 0;
    CgWorkspace& c = g_ctx.cg;
    dev_free(c.slab); dev_free(c.d_state); dev_free(c.d_trace);
    if (c.h_mirror) (void)hipHostFree(const_cast<unsigned long long*>(c.h_mirror));
    c = CgWorkspace();
    dev_free(g_ctx.scratch_dev); g_ctx.scratch_dev = nullptr;
    dev_free(g_ctx.rbuf); g_ctx.rbuf = nullptr; g_ctx.rbuf_cap = 0;
    if (g_ctx.mbox_h) { (void)hipHostFree(const_cast<unsigned long long*>(g_ctx.mbox_h)); g_ctx.mbox_h = nullptr; g_ctx.mbox_d = nullptr; }
    async_upload_destroy(g_ctx.upload_cache); g_ctx.upload_cache = nullptr;
    for (auto& im : g_ctx.image_pool) dev_free(im.ptr);
    g_ctx.image_pool.clear();
    if (g_ctx.own_stream) (void)hipStreamDestroy(g_ctx.own_stream);
    g_ctx.own_stream = nullptr; g_ctx.stream = nullptr;
    if (g_pin.base) { (void)hipHostFree(g_pin.base); g_pin = PinArena(); }
    g_ctx.vlds_attr_set = false; g_ctx.cgp_vlds_attr_set = false; g_ctx.trsm_lds_granted = false; g_ctx.trsv_lds_granted = 0;
    g_ctx.init = false; g_ctx.rank = 0; g_ctx.nranks = 1;
    return BH_OK;
}

int32_t bh_set_stream(void* hip_stream) {
    BH_REQUIRE_INIT();
    BH_TRY(sync_flush());
    g_ctx.stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : g_ctx.own_stream;
    return BH_OK;
}

int32_t bh_synchronize(void) {
    BH_REQUIRE_INIT();
    BH_TRY(sync_flush());
    return BH_OK;
}

int32_t bh_device_info(char* name_out, int64_t name_cap, int32_t* n_cu, char* arch_out, int64_t arch_cap) {
    BH_REQUIRE_INIT();
    hipDeviceProp_t prop;
    BH_HIP(hipGetDeviceProperties(&prop, g_ctx.device));
    if (name_out && name_cap > 0) { strncpy(name_out, prop.name, (size_t)name_cap - 1); name_out[name_cap - 1] = 0; }
    if (arch_out && arch_cap > 0) { strncpy(arch_out, prop.gcnArchName, (size_t)arch_cap - 1); arch_out[arch_cap - 1] = 0; }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    return BH_OK;
}

int32_t bh_set_option(const char* key, int64_t value) {
    if (!key) return fail(BH_ERR_INVALID_ARG, "NULL key");
    if (!strcmp(key, "rs_variant")) { g_ctx.opt_variant = value; return BH_OK; }
    if (!strcmp(key, "blocks_per_cu")) {
        if (value < 0 || value > kMaxBlocksPerCu) return fail(BH_ERR_INVALID_ARG, "blocks_per_cu must be 0 (default) .. 8");
        g_ctx.opt_blocks_per_cu = value;
        return BH_OK;
    }
    if (!strcmp(key, "pcg_batch")) { g_ctx.opt_batch = std::max<int64_t>(0, value); return BH_OK; }
    if (!strcmp(key, "pingpong")) { g_ctx.opt_pingpong = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "proj_form")) { g_ctx.opt_proj_form = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "fold_init")) { g_ctx.opt_fold_init = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "cg_fused")) {
        if (value < 0 || value > 2) return fail(BH_ERR_INVALID_ARG, "cg_fused is 0, 1 (box constraints) or 2 (also linear equalities)");
        g_ctx.opt_cg_fused = value;
        return BH_OK;
    }
    if (!strcmp(key, "final_sync")) { g_ctx.opt_final_sync = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "mailbox_flush")) { g_ctx.opt_mbox_flush = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "host_copy_kernels")) { g_ctx.opt_host_copy_kernels = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "ls_from_cg")) { g_ctx.opt_ls_from_cg = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "step_from_cg")) { g_ctx.opt_step_from_cg = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "gram_mfma")) { g_ctx.opt_gram_mfma = value; return BH_OK; }
    if (!strcmp(key, "chol_downdate")) { g_ctx.opt_chol_downdate = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "cauchy_image")) { g_ctx.opt_cauchy_image = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "cauchy_fused")) { g_ctx.opt_cauchy_fused = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "cauchy_fused_grid")) { g_ctx.opt_cauchy_fused_grid = std::min<int64_t>(std::max<int64_t>(0, value), kCauchyFusedGrid); return BH_OK; }
    if (!strcmp(key, "linv_refine")) { g_ctx.opt_linv_refine = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "cauchy_gemm")) { g_ctx.opt_cauchy_gemm = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "cauchy_image_max_ma")) { g_ctx.opt_cauchy_image_max_ma = std::min<int64_t>(std::max<int64_t>(0, value), 64); return BH_OK; }
    if (!strcmp(key, "chol_blocked")) { g_ctx.opt_chol_blocked = value ? 1 : 0; return BH_OK; }
    if (!strcmp(key, "image_pool")) {
        if (value < 0 || value > 8) return fail(BH_ERR_INVALID_ARG, "image_pool must be 0..8");
        g_ctx.opt_image_pool = value;
        while ((int64_t)g_ctx.image_pool.size() > value) { dev_free(g_ctx.image_pool.back().ptr); g_ctx.image_pool.pop_back(); }
        return BH_OK;
    }
    if (!strcmp(key, "upload_chunk_mb")) {
        if (value < 1 || value > 4096) return fail(BH_ERR_INVALID_ARG, "upload_chunk_mb must be 1..4096");
        g_ctx.opt_upload_chunk_mb = value;
        return BH_OK;
    }
    if (!strcmp(key, "profile_stride")) {
        if (value < 1) return fail(BH_ERR_INVALID_ARG, "profile_stride must be >= 1");
        g_ctx.opt_ev_stride = value;
        return BH_OK;
    }
    if (!strcmp(key, "comm_path")) {
        if (value == 1 && !g_ctx.peer.active) return fail(BH_ERR_PRECONDITION, "comm_path = 1 needs the peer-buffer communicator (BH_COMM=ipc or both)");
        if (value == 0 && comm_active() && g_ctx.comm == nullptr) return fail(BH_ERR_PRECONDITION, "comm_path = 0 needs an RCCL communicator (BH_COMM=rccl or both)");
        if (value != 0 && value != 1) return fail(BH_ERR_INVALID_ARG, "comm_path is 0 (RCCL) or 1 (peer buffers)");
        if (g_ctx.stream) (void)hipStreamSynchronize(g_ctx.stream);
        // leaving the peer-buffer path after a timed-out exchange: the error word is cleared so that the RCCL path can carry on
        // (re-selecting the peer path after a timeout is refused: the ranks' exchange counters no longer agree)
        if (value == 0 && g_ctx.peer.active && g_ctx.peer.h_err != nullptr && *g_ctx.peer.h_err != 0ull) {
            *g_ctx.peer.h_err = 0ull;
            g_ctx.peer.broken = true;
        }
        if (value == 1 && g_ctx.peer.broken) return fail(BH_ERR_RCCL, "the peer-buffer transport timed out earlier in this communicator's life");
        g_ctx.comm_path = (int)value;
        return BH_OK;
    }
    if (!strcmp(key, "profile")) { g_ctx.flags = value ? (g_ctx.flags | BH_FLAG_PROFILE) : (g_ctx.flags & ~BH_FLAG_PROFILE); return BH_OK; }
    return fail(BH_ERR_INVALID_ARG, std::string("unknown option ") + key);
}

// ---- multi-GPU -----------------------------------------------------------------------------
static int32_t load_rccl() {
    if (g_ctx.rccl_lib) return BH_OK;
    // BH_RCCL_LIB names a specific build.  Otherwise by SONAME: glibc hands back the copy the process has already mapped
    // (e.g. the one libtorch_hip.so brought in), so there is one RCCL per process.
    const char* names[] = {getenv("BH_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        if (!nm || !*nm) continue;
        g_ctx.rccl_lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (g_ctx.rccl_lib) break;
        if (nm == names[0]) return fail(BH_ERR_RCCL, std::string("dlopen(BH_RCCL_LIB=") + nm + "): " + dlerror());
    }
    if (!g_ctx.rccl_lib) return fail(BH_ERR_RCCL, std::string("dlopen(librccl): ") + dlerror());
#define BH_SYM(field, name)                                                          \
    g_ctx.field = reinterpret_cast<decltype(g_ctx.field)>(dlsym(g_ctx.rccl_lib, name)); \
    if (!g_ctx.field) return fail(BH_ERR_RCCL, std::string("dlsym ") + name)
    BH_SYM(p_ncclGetUniqueId, "ncclGetUniqueId");
    BH_SYM(p_ncclCommInitRank, "ncclCommInitRank");
    BH_SYM(p_ncclCommDestroy, "ncclCommDestroy");
    BH_SYM(p_ncclAllReduce, "ncclAllReduce");
    BH_SYM(p_ncclGetErrorString, "ncclGetErrorString");
#undef BH_SYM
    return BH_OK;
}

// Which communicators bh_comm_init brings up: BH_COMM = "rccl" (default), "ipc" (peer-buffer exchange only: no librccl
// needed) or "both" (RCCL carries the all-reduces until bh_set_option("comm_path", 1) switches to the peer buffers).
enum { COMM_RCCL = 1, COMM_IPC = 2 };
static int comm_mode() {
    const char* m = getenv("BH_COMM");
    if (!m || !*m || !strcmp(m, "rccl")) return COMM_RCCL;
    if (!strcmp(m, "ipc") || !strcmp(m, "peer")) return COMM_IPC;
    if (!strcmp(m, "both")) return COMM_RCCL | COMM_IPC;
    return 0;
}

static bool peer_barrier(PeerShm* shm, int nranks, int timeout_s) {
    const int gen = shm->generation.load(std::memory_order_acquire);
    if (shm->arrive.fetch_add(1, std::memory_order_acq_rel) + 1 == nranks) {
        shm->arrive.store(0, std::memory_order_relaxed);
        shm->generation.store(gen + 1, std::memory_order_release);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (shm->generation.load(std::memory_order_acquire) == gen) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) return false;
    }
    return true;
}

static void peer_release(bool with_barrier) {
    PeerComm& pc = g_ctx.peer;
    if (pc.shm && with_barrier) (void)peer_barrier(pc.shm, g_ctx.nranks, 20);   // nobody unmaps while a peer may still push
    for (int p = 0; p < kMaxPeers; ++p)
        if (pc.peer_base[p] && pc.peer_base[p] != pc.inbox) (void)hipIpcCloseMemHandle(pc.peer_base[p]);
    if (pc.inbox) (void)hipFree(pc.inbox);
    if (pc.h_err) (void)hipHostFree(const_cast<unsigned long long*>(pc.h_err));
    if (pc.shm) munmap(pc.shm, sizeof(PeerShm));
    pc = PeerComm();
}

// Bring up the peer-buffer exchange among the `nranks` processes of ONE node: allocate this rank's inbox, publish its
// hipIpc handle on a shared-memory page named after the communicator id, map everybody else's.
static int32_t peer_init(int rank, int nranks, const unsigned char* id) {
    if (nranks > kMaxPeers) return fail(BH_ERR_UNSUPPORTED, "peer-buffer all-reduce supports at most 8 ranks (one node)");
    PeerComm& pc = g_ctx.peer;
    unsigned long long h = 1469598103934665603ull;            // FNV-1a of the id: the rendezvous name
    for (int i = 0; i < BH_UNIQUE_ID_BYTES; ++i) { h ^= id[i]; h *= 1099511628211ull; }
    char name[64];
    snprintf(name, sizeof(name), "/bh_ipc_%016llx", h);
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return fail(BH_ERR_RCCL, std::string("shm_open(") + name + ") failed");
    if (ftruncate(fd, sizeof(PeerShm)) != 0) { close(fd); return fail(BH_ERR_RCCL, "ftruncate on the rendezvous page failed"); }
    void* m = mmap(nullptr, sizeof(PeerShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fail(BH_ERR_RCCL, "mmap of the rendezvous page failed");
    pc.shm = static_cast<PeerShm*>(m);

    // one allocation = one handle: [slots | flags | scal | seq, arrive, dead].  Fine-grained (uncached) so that a peer's stores and this
    // rank's polls meet in memory, not in somebody's L2.
    pc.inbox_bytes = kPeerSlotBytes + kPeerFlagBytes + kPeerScalBytes + 256;
    hipError_t e = hipExtMallocWithFlags(&pc.inbox, pc.inbox_bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&pc.inbox, pc.inbox_bytes, hipDeviceMallocFinegrained); }
    // no coarse-grained fallback: the exchange kernels issue no cache-invalidating acquire and no L2 write-back (their
    // system-scope accesses rely on memory that is not cached in a local L2), so on a plain hipMalloc'ed inbox flag polls and
    // payload loads could be served from a stale line.  The rendezvous fails instead (collectively) and callers use RCCL.
    if (e != hipSuccess) { (void)hipGetLastError(); pc.inbox = nullptr; }
    bool ok = (e == hipSuccess);
    if (ok) ok = hipMemset(pc.inbox, 0, pc.inbox_bytes) == hipSuccess;
    char* base = static_cast<char*>(pc.inbox);
    const unsigned long long one = 1ull;
    if (ok) ok = hipMemcpy(base + kPeerSlotBytes + kPeerFlagBytes + kPeerScalBytes, &one, sizeof(one), hipMemcpyHostToDevice) == hipSuccess;   // seq = 1
    if (ok) ok = hipDeviceSynchronize() == hipSuccess;
    if (ok && nranks > 1) ok = hipIpcGetMemHandle(&pc.shm->handle[rank], pc.inbox) == hipSuccess;
    if (!ok) pc.shm->failed = 1;
    if (!peer_barrier(pc.shm, nranks, 120)) { shm_unlink(name); peer_release(false); return fail(BH_ERR_RCCL, "peer rendezvous: the other ranks did not arrive (same node? same id?)"); }
    for (int p = 0; p < nranks && !pc.shm->failed; ++p) {
        if (p == rank) { pc.peer_base[p] = pc.inbox; continue; }
        if (hipIpcOpenMemHandle(&pc.peer_base[p], pc.shm->handle[p], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            pc.peer_base[p] = nullptr;
            pc.shm->failed = 1;
        }
    }
    const bool met = peer_barrier(pc.shm, nranks, 120);
    if (rank == 0 || !met) shm_unlink(name);   // everybody has mapped the page; nothing is left behind in /dev/shm (also on the failure exits)
    if (!met || pc.shm->failed) { peer_release(false); return fail(BH_ERR_RCCL, "peer rendezvous: hipIpc handle exchange failed on some rank"); }
    // from here on every failure is made COLLECTIVE (flag on the shared page + barrier) so that all ranks leave together
    void* hp = nullptr;
    void* dp = nullptr;
    bool local_ok = hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    if (local_ok) {
        memset(hp, 0, 64);
        pc.h_err = static_cast<volatile unsigned long long*>(hp);
        local_ok = hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess;
    }
    if (!local_ok) { (void)hipGetLastError(); pc.shm->failed = 1; }
    PeerArgs& a = pc.args;
    a = PeerArgs{};
    for (int p = 0; p < nranks; ++p) {
        char* b = static_cast<char*>(pc.peer_base[p]);
        a.slots[p] = reinterpret_cast<double*>(b);
        a.flags[p] = reinterpret_cast<unsigned long long*>(b + kPeerSlotBytes);
        a.scal[p] = reinterpret_cast<double*>(b + kPeerSlotBytes + kPeerFlagBytes);
    }
    a.seq = reinterpret_cast<unsigned long long*>(base + kPeerSlotBytes + kPeerFlagBytes + kPeerScalBytes);
    a.arrive = reinterpret_cast<unsigned*>(base + kPeerSlotBytes + kPeerFlagBytes + kPeerScalBytes + 64);
    a.dead = reinterpret_cast<unsigned*>(base + kPeerSlotBytes + kPeerFlagBytes + kPeerScalBytes + 128);
    a.err = static_cast<unsigned long long*>(dp);
    a.rank = rank; a.nranks = nranks; a.cap = kPeerCap; a.nblk_cap = kPeerBlkCap;
    const char* ts = getenv("BH_PEER_TIMEOUT_S");
    a.timeout_ticks = (unsigned long long)((ts && atof(ts) > 0 ? atof(ts) : 20.0) * 1e8);            // wall_clock64: 100 MHz
    // Reachability echo (a mapping that opens is not yet a mapping stores land in): one real exchange of a known vector
    // through every mapped inbox, with a short timeout, BEFORE anybody relies on the transport — a node where hipIpc maps
    // but peer stores do not arrive is found here, collectively, and callers fall back to RCCL instead of running into the
    // in-kernel timeout inside a timed region.  Rank r contributes (r + 1)(i + 1); the sum is exact in fp64.
    std::string echo_detail;
    const bool met1 = peer_barrier(pc.shm, nranks, 120);          // every rank's `failed` so far is visible behind it
    const char* skip = getenv("BH_PEER_ECHO_SKIP_RANK");          // test hook: this rank stays away from the echo
    if (met1 && !pc.shm->failed && skip && *skip && atoi(skip) == rank) {
        echo_detail = "this rank stayed away from the echo (BH_PEER_ECHO_SKIP_RANK)";
        pc.shm->failed = 1;
    } else if (met1 && !pc.shm->failed) {
        constexpr int kEcho = 2 * kPeerBlockChunks;               // one workgroup's worth
        double hv[kEcho], *dv = nullptr;
        for (int i = 0; i < kEcho; ++i) hv[i] = (double)(rank + 1) * (double)(i + 1);
        bool ok_e = hipMalloc(&dv, sizeof(hv)) == hipSuccess && hipMemcpy(dv, hv, sizeof(hv), hipMemcpyHostToDevice) == hipSuccess;
        if (ok_e) {
            PeerArgs ea = a;
            const char* es = getenv("BH_PEER_ECHO_TIMEOUT_S");
            ea.timeout_ticks = (unsigned long long)((es && atof(es) > 0 ? atof(es) : 5.0) * 1e8);
            hipLaunchKernelGGL(reduce_exchange_kernel, dim3(1), dim3(256), 0, g_ctx.stream, (const double*)nullptr, (int64_t)0, kPeerBlockChunks, 0, dv,
                               (const CgState*)nullptr, ea);
            ok_e = hipGetLastError() == hipSuccess && hipStreamSynchronize(g_ctx.stream) == hipSuccess &&
                   hipMemcpy(hv, dv, sizeof(hv), hipMemcpyDeviceToHost) == hipSuccess;
        }
        if (dv) (void)hipFree(dv);
        if (!ok_e) { (void)hipGetLastError(); echo_detail = "the echo exchange could not run"; }
        else if (*pc.h_err != 0ull) echo_detail = "the echo exchange timed out waiting for a peer's flag";
        else
            for (int i = 0; i < kEcho && echo_detail.empty(); ++i)
                if (hv[i] != 0.5 * nranks * (nranks + 1) * (i + 1)) echo_detail = "the echo exchange returned a wrong sum (a peer's stores did not land)";
        if (!echo_detail.empty()) pc.shm->failed = 1;
    } else {
        pc.shm->failed = 1;
    }
    const bool met2 = peer_barrier(pc.shm, nranks, 120);
    if (!met2 || pc.shm->failed) {
        peer_release(false);
        return fail(BH_ERR_RCCL, "peer rendezvous: reachability check failed" + (echo_detail.empty() ? std::string(" on another rank") : ": " + echo_detail));
    }
    pc.active = true;
    return BH_OK;
}

int32_t bh_comm_unique_id(void* id_out) {
    BH_REQUIRE_INIT();
    if (!id_out) return fail(BH_ERR_INVALID_ARG, "NULL id_out");
    static_assert(sizeof(ncclUniqueId) == BH_UNIQUE_ID_BYTES, "ncclUniqueId size");
    const int mode = comm_mode();
    if (mode == 0) return fail(BH_ERR_INVALID_ARG, "BH_COMM must be rccl, ipc or both");
    if (mode & COMM_RCCL) {
        BH_TRY(load_rccl());
        ncclUniqueId id;
        BH_NCCL(g_ctx.p_ncclGetUniqueId(&id));
        memcpy(id_out, &id, sizeof(id));
        return BH_OK;
    }
    // peer buffers only: the id merely names the rendezvous page
    unsigned char* o = static_cast<unsigned char*>(id_out);
    memset(o, 0, BH_UNIQUE_ID_BYTES);
    FILE* f = fopen("/dev/urandom", "rb");
    const size_t got = f ? fread(o, 1, 64, f) : 0;
    if (f) fclose(f);
    const unsigned long long salt = (unsigned long long)getpid() ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    memcpy(o + 64, &salt, sizeof(salt));
    memcpy(o + 72, &got, sizeof(got));
    return BH_OK;
}

int32_t bh_comm_init(int32_t rank, int32_t nranks, const void* id_in) {
    BH_REQUIRE_INIT();
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(BH_ERR_INVALID_ARG, "bad rank/nranks");
    if (comm_active()) return fail(BH_ERR_INVALID_ARG, "communicator already initialised");
    // nranks == 1 needs no communicator; BH_FORCE_COMM=1 creates a 1-rank communicator anyway so that the whole
    // exchange path (RCCL: dlopen, ncclCommInitRank, ncclAllReduce on the library stream; peer buffers: inbox, flags,
    // the fused reduce+exchange kernel) can be exercised and timed on a one-GPU box.
    const char* force = getenv("BH_FORCE_COMM");
    if (nranks == 1 && !(force && atoi(force) != 0)) { g_ctx.rank = 0; g_ctx.nranks = 1; return BH_OK; }
    // a bh_hess bakes in which rank applies the replicated C rows (rank 0): the rank must not change under a live handle
    if (g_ctx.live_hess > 0) return fail(BH_ERR_PRECONDITION, "bh_comm_init: destroy all bh_hess handles first (create them after the communicator)");
    if (!id_in) return fail(BH_ERR_INVALID_ARG, "NULL unique id");
    const int mode = comm_mode();
    if (mode == 0) return fail(BH_ERR_INVALID_ARG, "BH_COMM must be rccl, ipc or both");
    g_ctx.rank = rank;
    g_ctx.nranks = nranks;
    if (mode & COMM_RCCL) {
        int32_t rc = load_rccl();
        if (rc == BH_OK) {
            ncclUniqueId id;
            memcpy(&id, id_in, sizeof(id));
            ncclResult_t r = g_ctx.p_ncclCommInitRank(&g_ctx.comm, nranks, id, rank);
            if (r != ncclSuccess) { g_ctx.comm = nullptr; rc = fail(BH_ERR_RCCL, std::string("ncclCommInitRank: ") + g_ctx.p_ncclGetErrorString(r)); }
        }
        if (rc != BH_OK) { g_ctx.rank = 0; g_ctx.nranks = 1; return rc; }
    }
    if (mode & COMM_IPC) {
        const int32_t rc = peer_init(rank, nranks, static_cast<const unsigned char*>(id_in));
        if (rc != BH_OK) {
            if (g_ctx.comm && g_ctx.p_ncclCommDestroy) { g_ctx.p_ncclCommDestroy(g_ctx.comm); g_ctx.comm = nullptr; }
            g_ctx.rank = 0; g_ctx.nranks = 1;
            return rc;
        }
    }
    g_ctx.comm_path = (mode == COMM_IPC) ? 1 : 0;
    return BH_OK;
}

int32_t bh_comm_destroy(void) {
    if (comm_active() && g_ctx.live_hess > 0) return fail(BH_ERR_PRECONDITION, "bh_comm_destroy: destroy all bh_hess handles first");
    if (comm_active() && g_ctx.stream) (void)hipStreamSynchronize(g_ctx.stream);
    if (g_ctx.comm && g_ctx.p_ncclCommDestroy) g_ctx.p_ncclCommDestroy(g_ctx.comm);
    if (g_ctx.peer.active) peer_release(true);
    g_ctx.comm = nullptr; g_ctx.rank = 0; g_ctx.nranks = 1; g_ctx.comm_path = 0;
    return BH_OK;
}

int32_t bh_comm_info(int32_t* rank, int32_t* nranks) {
    if (rank) *rank = g_ctx.rank;
    if (nranks) *nranks = g_ctx.nranks;
    return BH_OK;
}

// ---- AlHessian ------------------------------------------------------------------------------
int32_t bh_hess_create(bh_hess** out, const double* J, int64_t d, int64_t n, int64_t ldJ, const double* C, int64_t q,
                       int64_t ldC, double mu) {
    BH_REQUIRE_INIT();
    if (!out) return fail(BH_ERR_INVALID_ARG, "NULL out");
    *out = nullptr;
    if (d < 0 || n < 1 || q < 0) return fail(BH_ERR_INVALID_ARG, "negative dimension");
    if (d > 0 && (!J || ldJ < d)) return fail(BH_ERR_INVALID_ARG, "J NULL or ldJ < d");
    if (q > 0 && (!C || ldC < q)) return fail(BH_ERR_INVALID_ARG, "C NULL or ldC < q");
    bh_hess* H = new bh_hess();
    H->d = d; H->n = n; H->q = q; H->mu = mu;
    int32_t rc = alloc_hess_common(H);
    if (rc == BH_OK) rc = upload_transposed(J, d, n, ldJ, H->Jd, 0, H->ld);
    if (rc == BH_OK) rc = upload_transposed(C, q, n, ldC, H->Jd, d, H->ld);
    if (rc == BH_OK) rc = finish_hess_create(H);
    if (rc == BH_OK) rc = sync_flush();
    if (rc != BH_OK) { bh_hess_destroy(H); return rc; }
    *out = H;
    H->counted = true;
    g_ctx.live_hess += 1;
    return BH_OK;
}

int32_t bh_hess_create_dev(bh_hess** out, const double* J_dev, int64_t d, int64_t n, int64_t ldJ, const double* C, int64_t q,
                           int64_t ldC, double mu) {
    BH_REQUIRE_INIT();
    if (!out) return fail(BH_ERR_INVALID_ARG, "NULL out");
    *out = nullptr;
    if (d < 0 || n < 1 || q < 0) return fail(BH_ERR_INVALID_ARG, "negative dimension");
    if (d > 0 && (!J_dev || ldJ < d)) return fail(BH_ERR_INVALID_ARG, "J NULL or ldJ < d");
    if (q > 0 && (!C || ldC < q)) return fail(BH_ERR_INVALID_ARG, "C NULL or ldC < q");
    bh_hess* H = new bh_hess();
    H->d = d; H->n = n; H->q = q; H->mu = mu;
    int32_t rc = alloc_hess_common(H);
    if (rc == BH_OK) rc = transpose_from_device(J_dev, d, n, ldJ, H->Jd, 0, H->ld);
    if (rc == BH_OK) rc = upload_transposed(C, q, n, ldC, H->Jd, d, H->ld);
    if (rc == BH_OK && hipStreamSynchronize(g_ctx.stream) != hipSuccess) rc = fail(BH_ERR_HIP, "bh_hess_create_dev: synchronize");
    if (rc == BH_OK) rc = finish_hess_create(H);
    if (rc != BH_OK) { bh_hess_destroy(H); return rc; }
    *out = H;
    H->counted = true;
    g_ctx.live_hess += 1;
    return BH_OK;
}

// f-4: the same upload without blocking the caller.  J must stay valid and unchanged until bh_hess_wait (or the first use
// of the handle, which waits implicitly) returns.
int32_t bh_hess_create_async(bh_hess** out, const double* J, int64_t d, int64_t n, int64_t ldJ, const double* C, int64_t q,
                             int64_t ldC, double mu) {
    BH_REQUIRE_INIT();
    if (!out) return fail(BH_ERR_INVALID_ARG, "NULL out");
    *out = nullptr;
    if (d < 0 || n < 1 || q < 0) return fail(BH_ERR_INVALID_ARG, "negative dimension");
    if (d > 0 && (!J || ldJ < d)) return fail(BH_ERR_INVALID_ARG, "J NULL or ldJ < d");
    if (q > 0 && (!C || ldC < q)) return fail(BH_ERR_INVALID_ARG, "C NULL or ldC < q");
    bh_hess* H = new bh_hess();
    H->d = d; H->n = n; H->q = q; H->mu = mu;
    int32_t rc = alloc_hess_common(H);
    if (rc == BH_OK) rc = upload_transposed(C, q, n, ldC, H->Jd, d, H->ld);      // the C block is small: synchronous
    if (rc == BH_OK && hipStreamSynchronize(g_ctx.stream) != hipSuccess) rc = fail(BH_ERR_HIP, "bh_hess_create_async: synchronize");
    if (rc == BH_OK && d > 0) {
        // column chunks of ~upload_chunk_mb MiB, a multiple of 32 columns (the transpose's tile width)
        int64_t cc = (g_ctx.opt_upload_chunk_mb << 20) / (8 * std::max<int64_t>(d, 1));
        cc = std::max<int64_t>(32, cc / 32 * 32);
        cc = std::min<int64_t>(cc, round_up(n, 32));
        const size_t need = (size_t)d * cc * sizeof(double);
        AsyncUpload* u = g_ctx.upload_cache;           // the previous upload's resources, if they are free and large enough
        g_ctx.upload_cache = nullptr;
        if (u && u->staging_bytes < need) { async_upload_destroy(u); u = nullptr; }
        bool ok = true;
        if (!u) {
            u = new AsyncUpload();
            ok = hipStreamCreateWithFlags(&u->s_copy, hipStreamNonBlocking) == hipSuccess &&
                 hipStreamCreateWithFlags(&u->s_xpose, hipStreamNonBlocking) == hipSuccess;
            for (int i = 0; i < 2 && ok; ++i) {
                ok = hipEventCreateWithFlags(&u->copied[i], hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&u->freed[i], hipEventDisableTiming) == hipSuccess &&
                     hipMalloc(reinterpret_cast<void**>(&u->staging[i]), need) == hipSuccess;
            }
            u->staging_bytes = need;
        }
        u->chunk_cols = cc;
        u->rc = BH_OK;
        if (!ok) {
            async_upload_destroy(u);
            rc = fail(BH_ERR_HIP, "bh_hess_create_async: streams / events / staging buffers");
        } else {
            H->up = u;
            u->worker = std::thread(async_upload_worker, H, J, d, n, ldJ, g_ctx.device);
        }
    } else if (rc == BH_OK) {
        // no rows on this rank (d_total < ranks): nothing to upload, but the d_total all-reduce of finish_hess_create must run where
        // the peers run theirs — at bh_hess_wait / first use — or any collective issued between create and wait (bh_resid_sqnorm,
        // another handle) would be matched against it
        if (comm_active()) H->pending_finish = true;
        else rc = finish_hess_create(H);
    }
    if (rc != BH_OK) { bh_hess_destroy(H); return rc; }
    *out = H;
    H->counted = true;
    g_ctx.live_hess += 1;
    return BH_OK;
}

int32_t bh_hess_wait(bh_hess* H) {
    BH_REQUIRE_INIT();
    if (!H) return fail(BH_ERR_INVALID_ARG, "NULL bh_hess");
    return hess_ready(H);
}

int32_t bh_hess_create_synthetic(bh_hess** out, int64_t d, int64_t n, int64_t row0, int64_t d_total, uint64_t seed,
                                 const double* colscale, double mu) {
    BH_REQUIRE_INIT();
    if (!out) return fail(BH_ERR_INVALID_ARG, "NULL out");
    *out = nullptr;
    if (d < 1 || n < 1 || row0 < 0 || d_total < row0 + d) return fail(BH_ERR_INVALID_ARG, "bad synthetic shape");
    bh_hess* H = new bh_hess();
    H->d = d; H->n = n; H->q = 0; H->mu = mu;
    int32_t rc = alloc_hess_common(H);
    if (rc != BH_OK) { bh_hess_destroy(H); return rc; }
    double* cs_dev = nullptr;
    if (colscale) {
        rc = dev_alloc(&cs_dev, n);
        if (rc == BH_OK && hipMemcpyAsync(cs_dev, colscale, (size_t)n * sizeof(double), hipMemcpyHostToDevice, g_ctx.stream) != hipSuccess)
            rc = fail(BH_ERR_HIP, "colscale upload");
        if (rc != BH_OK) { dev_free(cs_dev); bh_hess_destroy(H); return rc; }
    }
    hipLaunchKernelGGL(synth_fill_kernel, dim3(g_ctx.n_cu * 8), dim3(256), 0, g_ctx.stream, H->Jd, H->ld, d, n, row0, d_total,
                       seed, cs_dev, std::sqrt((double)d_total));
    hipError_t e = hipStreamSynchronize(g_ctx.stream);
    dev_free(cs_dev);
    if (e != hipSuccess) { bh_hess_destroy(H); return fail(BH_ERR_HIP, std::string("synth_fill: ") + hipGetErrorString(e)); }
    rc = finish_hess_create(H);
    if (rc != BH_OK) { bh_hess_destroy(H); return rc; }
    *out = H;
    H->counted = true;
    g_ctx.live_hess += 1;
    return BH_OK;
}

int32_t bh_hess_set_mu(bh_hess* H, double mu) {
    if (!H) return fail(BH_ERR_INVALID_ARG, "NULL bh_hess");
    H->mu = mu;
    return BH_OK;
}

int32_t bh_hess_destroy(bh_hess* H) {
    if (H && g_ctx.hw_note.H == H) g_ctx.hw_note = {};
    if (H && g_ctx.gm_note.H == H) g_ctx.gm_note = {};
    if (!H) return BH_OK;
    if (H->up) {                              // an upload still in flight: let it finish before its buffers go
        AsyncUpload* u = H->up;
        if (u->worker.joinable()) u->worker.join();
        async_upload_cleanup(u);
        H->up = nullptr;
    }
    if (H->counted) g_ctx.live_hess -= 1;
    if (g_ctx.init) (void)hipStreamSynchronize(g_ctx.stream);
    if (H->Jd && g_ctx.init && (int64_t)g_ctx.image_pool.size() < g_ctx.opt_image_pool && H->Jd_doubles >= (1 << 17)) {
        g_ctx.image_pool.push_back({H->Jd, H->Jd_doubles});       // >= 1 MiB images only: small ones are not worth a pool slot
    } else {
        dev_free(H->Jd);
    }
    dev_free(H->vpad); dev_free(H->zpad); dev_free(H->upad); dev_free(H->tbuf); dev_free(H->timg); dev_free(H->timg_gen);
    dev_free(H->partials); dev_free(H->sq_partials); dev_free(H->scalar);
    for (auto e : H->ev) if (e) (void)hipEventDestroy(e);
    delete H;
    return BH_OK;
}

int32_t bh_hess_shape(const bh_hess* H, int64_t* d, int64_t* n, int64_t* q) {
    if (!H) return fail(BH_ERR_INVALID_ARG, "NULL bh_hess");
    if (d) *d = H->d;
    if (n) *n = H->n;
    if (q) *q = H->q;
    return BH_OK;
}

static int32_t hmul_impl(bh_hess* H, const double* v, double* out, bool dev) {
    BH_REQUIRE_INIT();
    if (!H || !v || !out) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    const double* vp = H->vpad;
    if (dev) BH_TRY(device_operand(&vp, H->vpad, v, H->n, H->ld));
    else BH_TRY(stage_vec(H->vpad, v, H->n, false));
    BH_TRY(launch_hmul(H, vp, H->zpad, nullptr, -1));
    BH_TRY(fetch_vec(out, H->zpad, H->n, dev));
    BH_TRY(dev ? finish_device_call() : sync_flush());
    H->stats.n_hmul += 1;
    return BH_OK;
}
int32_t bh_hmul(bh_hess* H, const double* v, double* out_n) { return hmul_impl(H, v, out_n, false); }
int32_t bh_hmul_dev(bh_hess* H, const double* v_dev, double* out_n_dev) { return hmul_impl(H, v_dev, out_n_dev, true); }

int32_t bh_vthv(bh_hess* H, const double* v, double* out_scalar) {
    BH_REQUIRE_INIT();
    if (!H || !v || !out_scalar) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    BH_TRY(stage_vec(H->vpad, v, H->n, false));
    BH_TRY(launch_jv(H, H->vpad, nullptr, true, H->scalar));
    BH_TRY(allreduce_inplace(H->scalar, 1, H));
    BH_TRY(fetch_vec(out_scalar, H->scalar, 1, false));
    BH_TRY(sync_flush());
    H->stats.n_jv += 1;
    return BH_OK;
}

static int32_t jv_impl(bh_hess* H, const double* v, double* out, bool dev) {
    BH_REQUIRE_INIT();
    if (!H || !v || (!out && H->d > 0)) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    if (dev) {
        const double* vp = H->vpad;
        BH_TRY(device_operand(&vp, H->vpad, v, H->n, H->ld));
        BH_TRY(launch_jv(H, vp, out, false, nullptr));
        BH_TRY(finish_device_call());
    } else {
        BH_TRY(stage_vec(H->vpad, v, H->n, false));
        BH_TRY(launch_jv(H, H->vpad, H->upad, false, nullptr));
        BH_TRY(fetch_vec(out, H->upad, H->d, false));
        BH_TRY(sync_flush());
    }
    H->stats.n_jv += 1;
    return BH_OK;
}
int32_t bh_jv(bh_hess* H, const double* v, double* out_d) { return jv_impl(H, v, out_d, false); }
int32_t bh_jv_dev(bh_hess* H, const double* v_dev, double* out_d_dev) { return jv_impl(H, v_dev, out_d_dev, true); }

static int32_t jtv_impl(bh_hess* H, const double* u, double* out, bool dev) {
    BH_REQUIRE_INIT();
    if (!H || (!u && H->d > 0) || !out) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    const double* u_dev = u;
    if (!dev) { BH_TRY(stage_vec(H->upad, u, H->d, false)); u_dev = H->upad; }
    BH_TRY(launch_jtv(H, u_dev, H->zpad));
    BH_TRY(fetch_vec(out, H->zpad, H->n, dev));
    BH_TRY(dev ? finish_device_call() : sync_flush());
    H->stats.n_jtv += 1;
    return BH_OK;
}
int32_t bh_jtv(bh_hess* H, const double* u, double* out_n) { return jtv_impl(H, u, out_n, false); }
int32_t bh_jtv_dev(bh_hess* H, const double* u_dev, double* out_n_dev) { return jtv_impl(H, u_dev, out_n_dev, true); }

// ---- MixedConstraints -----------------------------------------------------------------------
int32_t bh_proj_create(bh_proj** out, const double* A, int64_t mA, int64_t n, int64_t ldA) {
    BH_REQUIRE_INIT();
    if (!out) return fail(BH_ERR_INVALID_ARG, "NULL out");
    *out = nullptr;
    if (mA < 0 || n < 1) return fail(BH_ERR_INVALID_ARG, "negative dimension");
    if (mA > 0 && (!A || ldA < mA)) return fail(BH_ERR_INVALID_ARG, "A NULL or ldA < mA");
    if (mA > n) return fail(BH_ERR_PRECONDITION, "mA > n");
    bh_proj* P = new bh_proj();
    P->mA = mA; P->n = n; P->ldA = round_up(n, 16);
    int32_t rc = dev_alloc(&P->Ad, std::max<int64_t>(mA, 1) * P->ldA);
    if (rc == BH_OK) rc = dev_alloc(&P->fixrank, P->ldA);
    if (rc == BH_OK) rc = dev_alloc(&P->fixidx, n);
    if (rc == BH_OK) rc = dev_alloc(&P->newidx, n);
    if (rc == BH_OK) rc = dev_alloc(&P->counts, 8);
    if (rc == BH_OK) rc = dev_alloc(&P->chunks_dev, (n + 63) / 64 + 1);
    if (rc == BH_OK) rc = dev_alloc(&P->tw, n + 16);
    if (rc == BH_OK) rc = dev_alloc(&P->rpad, P->ldA);
    if (rc == BH_OK) rc = dev_alloc(&P->vtmp, P->ldA);
    if (rc == BH_OK && hipMemsetAsync(P->rpad, 0, P->ldA * sizeof(double), g_ctx.stream) != hipSuccess) rc = fail(BH_ERR_HIP, "memset");
    if (rc == BH_OK && hipMemsetAsync(P->fixrank, 0xff, P->ldA * sizeof(int), g_ctx.stream) != hipSuccess) rc = fail(BH_ERR_HIP, "memset");
    if (rc == BH_OK) rc = upload_transposed(A, mA, n, ldA, P->Ad, 0, P->ldA);
    if (rc != BH_OK) { bh_proj_destroy(P); return rc; }
    P->nfix = 0; P->mpp = (int)mA;
    BH_TRY(sync_flush());
    *out = P;
    return BH_OK;
}

int32_t bh_proj_set_active(bh_proj* P, const uint64_t* fix_chunks, int64_t n, const double* L, int64_t mpp, int64_t ldL) {
    BH_REQUIRE_INIT();
    if (!P) return fail(BH_ERR_INVALID_ARG, "NULL bh_proj");
    if (n != P->n) return fail(BH_ERR_SHAPE, "fixvars length differs from n");
    int nfix = 0;
    if (fix_chunks) {
        for (int64_t wd = 0; wd < (n + 63) / 64; ++wd) {
            uint64_t bits = fix_chunks[wd];
            if (wd == (n + 63) / 64 - 1 && (n & 63)) bits &= (1ull << (n & 63)) - 1ull;      // stray bits beyond n do not count
            nfix += __builtin_popcountll(bits);
        }
    }
    const int64_t want = P->mA + nfix;
    const size_t nwords = (size_t)((n + 63) / 64);
    {   // The Julia shim re-pushes before every projection; with the reduced form (or no linear rows) the device state
        // depends on fixvars only, so an identical push is a no-op (no upload, no refactorisation).
        const bool form_reduced = (P->mA == 0) || (g_ctx.opt_proj_form != 0);
        if (form_reduced && P->active_set && P->reduced == (P->mA > 0) && fix_chunks && P->last_chunks.size() == nwords &&
            memcmp(P->last_chunks.data(), fix_chunks, nwords * sizeof(uint64_t)) == 0)
            return BH_OK;       // whole words compared: stray bits beyond n only ever cause a (harmless) rebuild
    }
    if (want > n) return fail(BH_ERR_PRECONDITION, "mpp = mA + count(fixvars) > n (src/polyhedral_constraints.jl:43,128)");
    const bool reduced = (P->mA > 0) && (g_ctx.opt_proj_form != 0);
    if (P->mA > 0) {
        if (!L && !reduced) return fail(BH_ERR_INVALID_ARG, "L is required when mA > 0 (augmented form)");
        P->have_L = false;
        if (L) {
            if (mpp != want) return fail(BH_ERR_SHAPE, "mpp != mA + count(fixvars)");
            if (ldL < mpp) return fail(BH_ERR_INVALID_ARG, "ldL < mpp");
            if (!reduced) {
                if (mpp * mpp > P->L_cap) {
                    dev_free(P->L); P->L = nullptr; P->L_cap = 0;
                    BH_TRY(dev_alloc(&P->L, mpp * mpp));
                    P->L_cap = mpp * mpp;
                }
                count_h2d((size_t)mpp * mpp * sizeof(double));
                note_dma();
                BH_HIP(hipMemcpy2DAsync(P->L, (size_t)mpp * sizeof(double), L, (size_t)ldL * sizeof(double), (size_t)mpp * sizeof(double),
                                        (size_t)mpp, hipMemcpyHostToDevice, g_ctx.stream));
                P->have_L = true;
            }
        }
        const size_t lds = trsv_lds_bytes(reduced ? (int)P->mA : (int)want);
        if (lds > kLdsPerCu) return fail(BH_ERR_UNSUPPORTED, "factor too large for the single-workgroup triangular solve");
        BH_TRY(ensure_trsv_lds(lds));
    } else if (mpp != want && L != nullptr) {
        return fail(BH_ERR_SHAPE, "mpp != count(fixvars) for mA == 0");
    }
    // The mask goes up as the BitVector image it arrived as (n/8 bytes, through the pinned arena and a copy-free kernel read);
    // two small kernels expand it into fixrank / fixidx on the device.  (Fallback without a mapped arena: expanded on the host.)
    double* pin = arena_kernels() ? pin_alloc((int64_t)nwords) : nullptr;
    if (pin != nullptr) {
        if (fix_chunks) memcpy(pin, fix_chunks, nwords * sizeof(uint64_t));
        else memset(pin, 0, nwords * sizeof(uint64_t));
        count_h2d(nwords * sizeof(uint64_t));
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((P->ldA + 255) / 256, 256));
        hipLaunchKernelGGL(flags_from_chunks_kernel, dim3(grid), dim3(256), 0, g_ctx.stream,
                           reinterpret_cast<const unsigned long long*>(pin_dev_view(pin)), P->fixrank, (int)n, (int)P->ldA);
        hipLaunchKernelGGL(canon_mask_kernel, dim3(1), dim3(CG_T), 0, g_ctx.stream, P->fixrank, P->fixidx, (int)n, (int)P->ldA,
                           (unsigned long long*)nullptr, P->counts);
        BH_HIP(hipGetLastError());
    } else {
        std::vector<int> rank((size_t)P->ldA, -1), idx;
        if (fix_chunks) {
            for (int64_t i = 0; i < n; ++i)
                if ((fix_chunks[i >> 6] >> (i & 63)) & 1ull) { rank[(size_t)i] = (int)idx.size(); idx.push_back((int)i); }
        }
        count_h2d((size_t)(P->ldA + nfix) * sizeof(int));
        note_dma();
        BH_HIP(hipMemcpyAsync(P->fixrank, rank.data(), (size_t)P->ldA * sizeof(int), hipMemcpyHostToDevice, g_ctx.stream));
        if (nfix > 0) BH_HIP(hipMemcpyAsync(P->fixidx, idx.data(), (size_t)nfix * sizeof(int), hipMemcpyHostToDevice, g_ctx.stream));
        BH_HIP(hipStreamSynchronize(g_ctx.stream));          // rank / idx are locals of this block
    }
    P->nfix = nfix; P->mpp = (int)want; P->reduced = reduced;
    int info_host = 0;
    if (reduced) {
        // reduced form: factor A_free A_free' (mA x mA) on the device; bound changes need no host factor
        BH_TRY(ensure_reduced_buffers(P));
        BH_TRY(launch_reduced_factor(P, nfix > 0, nullptr));
    }
    if (reduced && !g_pin.dma && g_ctx.mbox_h != nullptr) {
        // the factorisation flag rides to the host with the seal that ends the call
        BH_TRY(mbox_seal_and_wait(sizeof(int), nullptr, nullptr, nullptr, P->info, 1, mbox_dev<int>(kMbInts)));
        info_host = mbox_host<int>(kMbInts)[0];
    } else {
        if (reduced) {
            count_d2h(sizeof(int));
            note_dma();
            BH_HIP(hipMemcpyAsync(&info_host, P->info, sizeof(int), hipMemcpyDeviceToHost, g_ctx.stream));
        }
        BH_TRY(sync_flush());   // host vectors go out of scope
    }
    if (info_host != 0) {
        P->active_set = false;
        return fail(BH_ERR_PRECONDITION, "A_free*A_free' is not positive definite (PosDefException in the reference's cholesky)");
    }
    P->active_set = true;
    if (fix_chunks) {
        // whole words are kept and compared (Julia keeps the bits beyond n zero; stray bits only ever cause a rebuild)
        P->last_chunks.assign(fix_chunks, fix_chunks + nwords);
    } else {
        P->last_chunks.assign(nwords, 0ull);
    }
    return BH_OK;
}

int32_t bh_proj_destroy(bh_proj* P) {
    if (!P) return BH_OK;
    if (g_ctx.init) (void)hipStreamSynchronize(g_ctx.stream);
    dev_free(P->Ad); dev_free(P->fixrank); dev_free(P->fixidx); dev_free(P->L); dev_free(P->tw); dev_free(P->rpad); dev_free(P->vtmp);
    dev_free(P->Lr); dev_free(P->M); dev_free(P->W);            // (P->info points into P->counts)
    dev_free(P->newidx); dev_free(P->counts); dev_free(P->chunks_dev); dev_free(P->tpart);
    delete P;
    return BH_OK;
}

int32_t bh_proj_shape(const bh_proj* P, int64_t* mA, int64_t* n, int64_t* n_fixed) {
    if (!P) return fail(BH_ERR_INVALID_ARG, "NULL bh_proj");
    if (mA) *mA = P->mA;
    if (n) *n = P->n;
    if (n_fixed) *n_fixed = P->nfix;
    return BH_OK;
}

static int32_t project_impl(bh_proj* P, const double* r, double* v_out, bool dev) {
    BH_REQUIRE_INIT();
    BH_TRY(check_proj_ready(P));
    if (!r || !v_out) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    BH_TRY(stage_vec(P->rpad, r, P->n, dev));
    BH_TRY(launch_project(P, P->rpad, P->vtmp, nullptr));
    BH_TRY(fetch_vec(v_out, P->vtmp, P->n, dev));
    BH_TRY(dev ? finish_device_call() : sync_flush());
    return BH_OK;
}
int32_t bh_project(bh_proj* P, const double* r, double* v_out) { return project_impl(P, r, v_out, false); }
int32_t bh_project_dev(bh_proj* P, const double* r_dev, double* v_out_dev) { return project_impl(P, r_dev, v_out_dev, true); }

int32_t bh_left_mul(bh_proj* P, const double* x, double* out_mpp) {
    BH_REQUIRE_INIT();
    if (!P || !x || !out_mpp) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    BH_TRY(stage_vec(P->rpad, x, P->n, false));
    ProjArgs a = proj_args(P, nullptr, false);
    const int grid1 = a.mA + (a.nfix + 255) / 256;
    if (grid1 > 0) hipLaunchKernelGGL(proj_left_mul_kernel, dim3(grid1), dim3(256), 0, g_ctx.stream, a, P->rpad);
    BH_HIP(hipGetLastError());
    BH_TRY(fetch_vec(out_mpp, P->tw, P->mA + P->nfix, false));
    BH_TRY(sync_flush());
    return BH_OK;
}

int32_t bh_left_mul_tr(bh_proj* P, const double* y, double* out_n) {
    BH_REQUIRE_INIT();
    if (!P || !y || !out_n) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    BH_TRY(stage_vec(P->tw, y, P->mA + P->nfix, false));
    ProjArgs a = proj_args(P, nullptr, false);
    const int nch = ((int)P->n + 1) / 2;
    hipLaunchKernelGGL((proj_left_mul_tr_kernel<false, 4>), dim3((nch + 63) / 64), dim3(256), 0, g_ctx.stream, a,
                       (const double*)nullptr, P->vtmp);
    BH_HIP(hipGetLastError());
    BH_TRY(fetch_vec(out_n, P->vtmp, P->n, false));
    BH_TRY(sync_flush());
    return BH_OK;
}

// ---- projected_cg ---------------------------------------------------------------------------
// results_final: every kernel that writes w (or H*w) had COMPLETED before the launch that published `done` started (two-kernel
// iteration, loop stopped by the exit test in an H*p launch's prologue: solved / iterations exhausted).  The device-pointer entry
// point then returns without draining the stream: what is still in flight are prologue-only launches that write nothing.
struct PcgFin { int done, status, iter, n_hmul; bool results_final = false; unsigned tag = 0; };

// How many iterations the host enqueues per launch-ahead batch.  One batch is always in flight while the host waits for
// the one before it, so the GPU never idles as long as the host can enqueue a batch faster than the GPU runs one; every
// iteration enqueued past the exit is a gated no-op that still costs ~1.5 us per kernel (measured, tools/dispatch_floor.py)
// and, with several ranks, a full all-reduce.  Long iterations (large J) therefore want batch 1, short ones a deeper queue.
// Measured at config 3 "ic" (23 iterations of 315 us): 7.247 ms per subproblem at batch 1, 7.281 at 4, 7.775 at 64.
static int launch_batch_size(const bh_hess* H) {
    if (g_ctx.opt_batch > 0) return (int)g_ctx.opt_batch;
    // rank-independent on purpose (see finish_hess_create): the even share of the rows over all ranks plus the full C
    // block, not this rank's own d + q_eff
    const double rows = (double)((H->d_total + g_ctx.nranks - 1) / g_ctx.nranks + H->q);
    const double est_us = (multi_panel(H) ? 16.0 : 8.0) * rows * (double)H->n / 7.0e6;      // ~7 TB/s streaming rate
    return est_us >= 100.0 ? 1 : est_us >= 40.0 ? 2 : 4;
}
// The first batch is sized by the previous call on the handle (consecutive subproblems of a minor loop behave alike):
// an exact prediction means no gated launches and no host round trip inside the loop at all.
constexpr int kFirstBatchCap = 32;
constexpr int kDowndateRefresh = 8;     // bh_cauchy_step, chol_downdate = 1: breakpoints between two from-scratch factorisations

// Launches the whole projected_cg on device vectors and returns once the host has seen the loop finish (the stream may
// still hold over-launched no-op kernels).  gp/wlp/wup: n doubles readable in 16-byte chunks; wp: output.
static int32_t pcg_run(bh_hess* H, bh_proj* P, const double* gp, const double* wlp, const double* wup, double* wp, bool w_in_ws,
                       double kappa2, double atol_negcurv, double atol_f2b, int64_t trace_cap, PcgFin* fin_out, double* hw = nullptr,
                       bool g_pad_zeroed = false /* the staged copy of g arrived with its padding already zero */) {
    const int64_t n = H->n, n_pad = H->ld;
    const int64_t max_iter64 = 2 * (n - P->mA - P->nfix);   // src/basic_tralcnlss.jl:714
    if (max_iter64 < 0) return fail(BH_ERR_PRECONDITION, "n - mA - count(fixvars) < 0");
    if (max_iter64 >= 0xfffff) return fail(BH_ERR_UNSUPPORTED, "2*(n - mA - count(fixvars)) exceeds the 20-bit iteration counters of the progress word");
    const int max_iter = (int)max_iter64;
    CgWorkspace& c = g_ctx.cg;
    H->ev_pending.clear();
    hipStream_t s = g_ctx.stream;
    const bool in_place = !w_in_ws;
    if (hw != nullptr) g_ctx.hw_note = {};          // cg.hw is about to be overwritten

    CgArgs a{};
    a.st = c.d_state; a.w = wp; a.r = c.r; a.v = c.v; a.p = c.p; a.Hp = c.Hp; a.g = gp; a.wl = wlp; a.wu = wup;
    a.fixrank = P->nfix > 0 ? P->fixrank : nullptr;
    a.n = (int)n; a.n_pad = (int)n_pad; a.w_in_ws = in_place ? 0 : 1; a.max_iter = max_iter; a.kappa2 = kappa2; a.atol_neg = atol_negcurv; a.atol_f2b = atol_f2b;
    a.trace = trace_cap > 0 ? c.d_trace : nullptr; a.trace_cap = (int)std::min<int64_t>(trace_cap, 0x7fffffff);
    c.tag = (c.tag + 1) & 0xffffu;
    if (c.tag == 0) c.tag = 1;
    a.mirror = c.d_mirror; a.tag = c.tag;
    a.hw = hw;

    const bool box = (P->mA == 0);
    // Box constraints, one rank, J rows register-resident: TWO kernels per iteration instead of three (bh_cgfuse.hip.h) — the
    // H*p launch forms p on the fly and takes the exit test, one 128-workgroup kernel reduces the slabs and updates w, r, v.
    // (g doubles as the first H*p input, so it must be readable up to the padded length: workspace copy or n == ld.)
    const int rs_cfg = multi_panel(H) ? -1 : pick_config(H->nchunks);
    // General constraints in the reduced projection form with mA <= 64 get the same treatment.
    // Default (cg_fused = 1): THREE kernels instead of seven — H*p (p formed on the fly), reduce/update leaving partials of A_free r,
    // and proj_apply_linv_kernel, whose every workgroup sums those partials and applies the explicit inverse of the factor
    // (tri_inv_small_kernel, rebuilt only when the factor changed: two 64-term dot products per entry instead of the
    // single-workgroup 128-step triangular solve) before forming its chunks of v = P(r) and its partial of r.v.
    // cg_fused = 2: FOUR kernels — the triangular solves in a launch of their own (trsv_small_kernel summing the partials),
    // then left_mul_tr.  Both change pHp's rounding like the box form does (cg_fused = 0 keeps dot(p, H*p)).
    const bool fuse_gen = !box && g_ctx.opt_cg_fused >= 1 && P->reduced && P->mA <= 64 && P->tpart != nullptr && P->ldA == H->ld;
    const bool gen_linv = fuse_gen && g_ctx.opt_cg_fused == 1 && P->W != nullptr;
    // (one equality: the "factor" is a scalar, y = t / l^2 has no conditioning to repair — the refinement step would only move the last bit)
    const bool linv_refine = gen_linv && g_ctx.opt_linv_refine != 0 && P->M_valid && P->mA >= 2;
    // Several ranks: the two-kernel (equalities: three-kernel) form carries the exchange inside the update kernel when the
    // peer-buffer transport is the active one (cg_reduce_update_kernel<GEN, true>: push the workgroup's 32 columns + its rank's share of pHp, wait, sum in
    // rank order); an RCCL all-reduce cannot sit inside a kernel, so that path keeps the three-kernel form.
    const bool peer_fused = comm_active() && use_peer_path() && (box || fuse_gen) && (H->nchunks + 15) / 16 <= kPeerBlkCap;
    // Equalities over RCCL: the collective is enqueued by the host between the slab reduction (which packs this rank's share of
    // p'Hp behind the vector, as in the box form below) and the update kernel, which then sums ONE slab — the all-reduced H*p:
    //   S(1) | R(1) AR(1) U(1) P(1) S(2) | ...   four kernels + the collective instead of seven, on the lock-step launch schedule.
    const bool rccl_gen = comm_active() && !use_peer_path() && fuse_gen;
    if ((box || fuse_gen) && g_ctx.opt_cg_fused && rs_cfg >= 0 && cgp_supported(rs_cfg) && (!comm_active() || peer_fused || rccl_gen) && max_iter >= 1 &&
        (gp == c.g || n == n_pad)) {
        BH_TRY(hess_ready(H));
        H->stats.cg_kernels = (box ? 2 : (gen_linv ? 3 : 4)) + (rccl_gen ? 1 : 0);
        if (gp == c.g && n < n_pad && !g_pad_zeroed) BH_HIP(hipMemsetAsync(c.g + n, 0, (size_t)(n_pad - n) * sizeof(double), s));
        const int64_t nrows = H->d + H->q_eff;
        const int grid = grid_for(rs_cfg, nrows);
        const int nblk = (H->nchunks + 15) / 16;
        double* pbuf[2] = {c.p, c.p2};
        double* rvbuf[2] = {c.rvpart, c.rvpart + n_pad / 2};
        const int nch_v = gen_linv ? H->nchunks : ((int)n + 1) / 2;    // proj_apply_linv_kernel also keeps the padding of v at zero
        const int nrv = fuse_gen ? (nch_v + 63) / 64 : nblk;            // who writes the r.v partials: the projection kernel (64 chunks per workgroup) or the update kernel
        const int vv_off = (int)(n_pad / 4);                            // v.v partials of the init launch, behind the r.v partials (nrv <= n_pad / 128)
        if (gen_linv) {
            if (!P->linv_valid) {
                hipLaunchKernelGGL(tri_inv_small_kernel, dim3(1), dim3(256), 0, s, (const double*)P->Lr, (int)P->mA, P->W);
                P->linv_valid = true;
            }
            // :705-710 in two launches: tw = A_free mask(g); v = P(g), p_1 = -v and the partials of r.v, v.v.  r = g and w = 0 are
            // formed by the first update kernel, CgState is set by workgroup 0 of the first H*p launch (init_done = 2).
            ProjArgs pa = proj_args(P, nullptr, true);
            hipLaunchKernelGGL(proj_left_mul_kernel, dim3((unsigned)P->mA), dim3(256), 0, s, pa, gp);
            pa.tpart = P->tw; pa.tpart_nblk = 1; pa.rvpart = rvbuf[0]; pa.vvpart = rvbuf[0] + vv_off; pa.p_out = c.p;
            pa.W = P->W; pa.nch_pad = H->nchunks; pa.Mgram = linv_refine ? P->M : nullptr; pa.fused_j = 0;
            hipLaunchKernelGGL((proj_apply_linv_kernel<true>), dim3((unsigned)nrv), dim3(256), 0, s, pa, gp, c.v);
        } else if (fuse_gen) {
            // :702-718 by the init kernels: r = g, w = 0, v = P(r), rtv, p_1 = -v (in c.p), tol_cg, CgState (stop_at = 0)
            hipLaunchKernelGGL((cg_init_kernel<false>), dim3(1), dim3(CG_T), 0, s, a);
            BH_TRY(launch_project(P, c.r, c.v, c.d_state));
            hipLaunchKernelGGL(cg_init_finish_kernel, dim3(1), dim3(CG_T), 0, s, a);
        }
        int expect_stop_at = H->last_n_hmul > 0 ? H->last_n_hmul + 1 : 0;     // the launch expected to find the loop finished
        auto launch_stream = [&](int j) -> int32_t {            // H*p of iteration j (1-based), p_j formed on the fly
            RowStreamArgs ra = rs_args(H, nrows, nullptr);
            ra.partials = H->partials;
            if (fuse_gen) { ra.v = c.p; ra.negate = 0; }                            // used by j == 1 only: p_1 = -P(g), formed by the init kernels
            else { ra.v = gp; ra.negate = 1; ra.negmask = a.fixrank; }             //                      p_1 = -mask(g), formed on the fly
            CgFuse& f = ra.cf;
            f.st = c.d_state; f.j = j; f.n = (int)n; f.max_iter = max_iter; f.init_done = gen_linv ? 2 : fuse_gen ? 1 : 0; f.vv_off = vv_off;
            f.vvec = c.v; f.p_old = pbuf[(j - 1) & 1]; f.p_new = pbuf[j & 1];
            f.rvpart = rvbuf[(j - 1) & 1]; f.nrv = nrv;
            f.w = wp; f.wl = wlp; f.wu = wup;
            f.sqpart = H->sq_partials; f.gpart = H->sq_partials + H->g_cap;      // one entry per WORKGROUP (up to g_cap of them), not per chunk
            f.kappa2 = kappa2; f.atol_f2b = atol_f2b;
            f.trace = a.trace; f.trace_cap = a.trace_cap; f.mirror = a.mirror; f.tag = a.tag;
            int slot = -1;
            const bool expect_stop = j == expect_stop_at || j > max_iter || (expect_stop_at == 0 && j > 1);
            BH_TRY(profile_begin(H, j - 1, &slot));
            // a handle without history cannot predict its exit: its look-ahead launches (j > 1) take the no-prefetch symbol too, so
            // that launches which stop in their prologue never show up under the streaming kernel's name in a profile
            launch_row_stream_cgp(rs_cfg, ra, grid, s, expect_stop);
            if (slot >= 0) BH_HIP(hipEventRecord(H->ev[2 * slot + 1], s));
            return BH_OK;
        };
        auto launch_update = [&](int j) -> int32_t {
            CgUpdArgs u{};
            u.st = c.d_state; u.j = j; u.partials = H->partials; u.ld = H->ld; u.nchunks = H->nchunks; u.G = grid; u.Gs = grid; u.Gq = grid;
            u.sqpart = H->sq_partials; u.gpart = H->sq_partials + H->g_cap; u.rvpart_in = rvbuf[(j - 1) & 1]; u.rvpart_out = rvbuf[j & 1]; u.nrv = nrv;
            u.p = pbuf[j & 1]; u.w = wp; u.hw = hw; u.r = c.r; u.g = gp; u.v = c.v; u.fixrank = a.fixrank;
            u.n = (int)n; u.atol_neg = atol_negcurv; u.trace = a.trace; u.trace_cap = a.trace_cap; u.mirror = a.mirror; u.tag = a.tag;
            if (!fuse_gen) {
                if (peer_fused) hipLaunchKernelGGL((cg_reduce_update_kernel<false, true>), dim3(nblk), dim3(256), 0, s, u, g_ctx.peer.args);
                else hipLaunchKernelGGL((cg_reduce_update_kernel<false, false>), dim3(nblk), dim3(256), 0, s, u, PeerArgs{});
                return BH_OK;
            }
            if (rccl_gen) {
                hipLaunchKernelGGL(reduce_partials_sq_kernel, dim3(nblk), dim3(256), 0, s, (const double*)H->partials, H->ld, H->nchunks, grid,
                                   c.hpx, (const double*)H->sq_partials, (int)n_pad, (const CgState*)c.d_state, j);
                BH_HIP(hipGetLastError());
                BH_TRY(allreduce_inplace(c.hpx, n_pad + 2, H));
                u.partials = c.hpx; u.Gs = 1; u.sqpart = c.hpx + n_pad; u.Gq = 1;
            }
            // (A, L and the vectors are replicated: past the exchange every rank forms the same partials of A_free r, the same v)
            u.A = P->Ad; u.ldA = P->ldA; u.mA = (int)P->mA; u.tpart = P->tpart; u.init_in_memory = gen_linv ? 0 : 1;
            if (peer_fused) hipLaunchKernelGGL((cg_reduce_update_kernel<true, true>), dim3(nblk), dim3(256), 0, s, u, g_ctx.peer.args);
            else hipLaunchKernelGGL((cg_reduce_update_kernel<true, false>), dim3(nblk), dim3(256), 0, s, u, PeerArgs{});
            // y = (A_free A_free')^{-1} (A_free r): partial sums + the two triangular solves; then v = r_free - A_free'y and r.v
            ProjArgs pa = proj_args(P, c.d_state, true);
            pa.tpart = P->tpart; pa.tpart_nblk = nblk; pa.rvpart = rvbuf[j & 1]; pa.fused_j = j;
            if (gen_linv) {
                pa.W = P->W; pa.nch_pad = H->nchunks; pa.Mgram = linv_refine ? P->M : nullptr;
                hipLaunchKernelGGL((proj_apply_linv_kernel<false>), dim3((unsigned)nrv), dim3(256), 0, s, pa, (const double*)c.r, c.v);
                return BH_OK;
            }
            hipLaunchKernelGGL(trsv_small_kernel, dim3(1), dim3(256), 0, s, pa);
            hipLaunchKernelGGL((proj_left_mul_tr_kernel<true, 4>), dim3((nch_v + 63) / 64), dim3(256), 0, s, pa, (const double*)c.r, c.v);
            return BH_OK;
        };
        // Launch order: S(1) | U(1) S(2) | U(2) S(3) | ...: the stream kernel of iteration j+1 is what detects "solved" after
        // iteration j, so it is always enqueued together with U(j) (as a gated no-op if the loop ended in U(j)).
        BH_TRY(launch_stream(1));
        int launched = 0;                                          // iterations whose U has been enqueued
        auto launch_batch = [&](int nb) -> int32_t {
            nb = std::min(nb, max_iter - launched);
            for (int i = 0; i < nb; ++i) {
                BH_TRY(launch_update(launched + 1));
                BH_TRY(launch_stream(launched + 2));               // (iteration max_iter + 1 only ever runs its prologue: iter > max_iter)
                launched += 1;
            }
            BH_HIP(hipGetLastError());
            return BH_OK;
        };
        const int batch = launch_batch_size(H);
        MirrorWord mw{};
        constexpr int kNever = 0x7fffffff;
        if (rccl_gen) {
            // every enqueued iteration holds a collective: all ranks must take the same launch decisions, so a decision taken after
            // S(target + 1) has spoken uses "the loop had ended by iteration `target`" only, never a later state one rank happened
            // to see (the lock-step schedule of the separate-kernel form, on this form's progress protocol)
            auto done_by = [&](int target) { return mw.done && mw.n_hmul <= target; };
            BH_TRY(launch_batch(H->last_n_hmul > 0 ? std::min(H->last_n_hmul, kFirstBatchCap) : 1));
            BH_TRY(wait_mirror(c, a.tag, kNever, &mw, launched + 1));
            if (!done_by(launched) && launched < max_iter) {
                BH_TRY(launch_batch(batch));
                while (true) {
                    const int target = launched;
                    const bool more = launched < max_iter;
                    if (more) BH_TRY(launch_batch(batch));
                    BH_TRY(wait_mirror(c, a.tag, kNever, &mw, target + 1));
                    if (done_by(target) || !more) break;
                }
            }
            BH_TRY(wait_mirror(c, a.tag, kNever, &mw));
            if (!mw.done) return fail(BH_ERR_HIP, "internal: CG loop did not terminate");
            fin_out->done = mw.done; fin_out->status = mw.status; fin_out->iter = mw.iter; fin_out->n_hmul = mw.n_hmul;
            fin_out->tag = a.tag;             // (results_final stays false: over-launched collectives are still in flight)
            return BH_OK;
        }
        const int first = H->last_n_hmul > 0 ? std::min(H->last_n_hmul, kFirstBatchCap) : std::min(batch, 2);
        BH_TRY(launch_batch(first));
        // S(launched + 1) speaks for iteration `launched`: either "stopped" (done) or "iter = launched + 1, streaming"
        BH_TRY(wait_mirror(c, a.tag, kNever, &mw, launched + 1));
        if (!mw.done && launched < max_iter) {
            BH_TRY(launch_batch(batch));
            while (true) {
                const int target = launched;             // everything enqueued so far except the batch launched next
                const bool more = launched < max_iter;
                if (more) BH_TRY(launch_batch(batch));
                BH_TRY(wait_mirror(c, a.tag, kNever, &mw, target + 1));
                if (mw.done || !more) break;
            }
        }
        BH_TRY(wait_mirror(c, a.tag, kNever, &mw));          // the final state
        if (!mw.done) return fail(BH_ERR_HIP, "internal: CG loop did not terminate");
        fin_out->done = mw.done; fin_out->status = mw.status; fin_out->iter = mw.iter; fin_out->n_hmul = mw.n_hmul;
        fin_out->results_final = (mw.status == BH_CG_SOLVED || mw.status == BH_CG_MAX_ITER_REACHED || mw.status == BH_CG_NONE);
        fin_out->tag = a.tag;
        // exchanges that really ran (one per executed H*p): how many update kernels were ENQUEUED past the exit depends on when
        // this rank polled its progress word, and those exchange nothing
        if (peer_fused) H->stats.n_allreduce += mw.n_hmul;
        return BH_OK;
    }
    // Several ranks over RCCL, box constraints: TWO kernels and the collective per iteration.  An ncclAllReduce is
    // enqueued by the host, so it cannot sit inside the update kernel as the peer-buffer exchange does; instead the update of
    // iteration j-1 moves into the prologue of the H*p launch of iteration j (row_stream_kernel<..., CGP = 3>: every workgroup
    // redoes it for the whole 32 KiB vector, the owners of a chunk store it), and the slab reduction packs this rank's share of
    // p'Hp behind the vector that goes through the all-reduce:  S(1) | R(1) AR(1) S(2) | R(2) AR(2) S(3) | ...
    // The launch schedule is the lock-step one of the three-kernel form below (decisions on "done by iteration k" only).
    const bool rccl_fused = comm_active() && !use_peer_path() && box && g_ctx.opt_cg_fused && rs_cfg >= 0 && cgp3_supported(rs_cfg) &&
                            max_iter >= 1 && (gp == c.g || n == n_pad);
    H->stats.cg_kernels = 0;
    if (rccl_fused) {
        BH_TRY(hess_ready(H));
        H->stats.cg_kernels = 2;
        if (gp == c.g && n < n_pad && !g_pad_zeroed) BH_HIP(hipMemsetAsync(c.g + n, 0, (size_t)(n_pad - n) * sizeof(double), s));
        const int64_t nrows = H->d + H->q_eff;
        const int grid = grid_for(rs_cfg, nrows);
        const int nblk = (H->nchunks + 15) / 16;
        double* pbuf[2] = {c.p, c.p2};
        double* rbuf[2] = {c.r, c.r2};
        double* gbuf[2] = {H->sq_partials + H->g_cap, H->sq_partials + 2 * (int64_t)H->g_cap};       // gamma partials, one per workgroup
        auto launch_stream = [&](int j) -> int32_t {
            RowStreamArgs ra = rs_args(H, nrows, nullptr);
            ra.partials = H->partials;
            ra.v = gp; ra.negate = 1; ra.negmask = a.fixrank;                // j == 1: p_1 = -mask(g), formed on the fly
            CgFuse& f = ra.cf;
            f.st = c.d_state; f.j = j; f.n = (int)n; f.max_iter = max_iter; f.init_done = 0;
            f.p_old = pbuf[(j - 1) & 1]; f.p_new = pbuf[j & 1];
            f.r_old = rbuf[(j - 1) & 1]; f.r_new = rbuf[j & 1];
            f.gpart_in = gbuf[(j - 1) & 1]; f.gpart = gbuf[j & 1];
            f.w_rw = wp; f.w = wp; f.hw = hw; f.g = gp; f.fixrank = a.fixrank;
            f.wl = wlp; f.wu = wup;
            f.sqpart = H->sq_partials;
            f.Hp = c.hpx; f.sq_index = (int)n_pad;
            f.kappa2 = kappa2; f.atol_f2b = atol_f2b; f.atol_neg = atol_negcurv;
            f.trace = a.trace; f.trace_cap = a.trace_cap; f.mirror = a.mirror; f.tag = a.tag;
            int slot = -1;
            BH_TRY(profile_begin(H, j - 1, &slot));
            launch_row_stream_cgp3(rs_cfg, ra, grid, s);
            if (slot >= 0) BH_HIP(hipEventRecord(H->ev[2 * slot + 1], s));
            return BH_OK;
        };
        auto launch_iteration = [&](int index) -> int32_t {                 // index = j - 1: R(j), AR(j), S(j + 1)
            const int j = index + 1;
            hipLaunchKernelGGL(reduce_partials_sq_kernel, dim3(nblk), dim3(256), 0, s, (const double*)H->partials, H->ld, H->nchunks, grid,
                               c.hpx, (const double*)H->sq_partials, (int)n_pad, (const CgState*)c.d_state, j);
            BH_HIP(hipGetLastError());
            BH_TRY(allreduce_inplace(c.hpx, n_pad + 2, H));
            return launch_stream(j + 1);
        };
        BH_TRY(launch_stream(1));
        const int batch = launch_batch_size(H);
        int launched = 0;
        auto launch_batch = [&](int nb) -> int32_t {
            nb = std::min(nb, max_iter - launched);
            for (int i = 0; i < nb; ++i) BH_TRY(launch_iteration(launched + i));
            launched += nb;
            return BH_OK;
        };
        MirrorWord mw{};
        auto done_by = [&](int target) { return mw.done && mw.n_hmul <= target; };
        const int first = H->last_n_hmul > 0 ? std::min(H->last_n_hmul, kFirstBatchCap) : 1;
        BH_TRY(launch_batch(first));
        BH_TRY(wait_mirror(c, a.tag, launched, &mw));
        if (!done_by(launched) && launched < max_iter) {
            BH_TRY(launch_batch(batch));
            while (true) {
                const int target = launched;
                const bool more = launched < max_iter;
                if (more) BH_TRY(launch_batch(batch));
                BH_TRY(wait_mirror(c, a.tag, target, &mw));
                if (done_by(target) || !more) break;
            }
        }
        BH_TRY(wait_mirror(c, a.tag, launched, &mw));
        if (!mw.done) return fail(BH_ERR_HIP, "internal: CG loop did not terminate");
        fin_out->done = mw.done; fin_out->status = mw.status; fin_out->iter = mw.iter; fin_out->n_hmul = mw.n_hmul;
        fin_out->tag = a.tag;
        return BH_OK;
    }
    // Box constraints with register-resident vectors: no init kernel — the first H*p forms p0 = -mask(g) on the fly and
    // the first step kernel does the initialisation of :702-718 itself.
    // (g doubles as the H*p input there, so it must be readable up to the padded length: workspace copy or n == ld.)
    const bool fold_init = box && !multi_panel(H) && ((n + 1) / 2 <= 4 * CG_T) && max_iter >= 1 && g_ctx.opt_fold_init &&
                           (gp == c.g || n == n_pad);
    if (fold_init) {
        if (gp == c.g && n < n_pad && !g_pad_zeroed)   // stale padding of the staged g would be multiplied into the dot products
            BH_HIP(hipMemsetAsync(c.g + n, 0, (size_t)(n_pad - n) * sizeof(double), s));
    } else if (box) {
        hipLaunchKernelGGL((cg_init_kernel<true>), dim3(1), dim3(CG_T), 0, s, a);
    } else {
        hipLaunchKernelGGL((cg_init_kernel<false>), dim3(1), dim3(CG_T), 0, s, a);
        BH_TRY(launch_project(P, c.r, c.v, c.d_state));
        hipLaunchKernelGGL(cg_init_finish_kernel, dim3(1), dim3(CG_T), 0, s, a);
    }
    BH_HIP(hipGetLastError());

    auto launch_iteration = [&](int index) -> int32_t {
        if (fold_init && index == 0) {
            // the state still holds the previous call's `done`: this launch is never a no-op, so it is not gated
            BH_TRY(launch_hmul(H, gp, c.Hp, nullptr, index, 0, true, a.fixrank));                 // :722 with p = -P(g)
            launch_cg_first_step(a, s);
            return BH_OK;
        }
        BH_TRY(launch_hmul(H, c.p, c.Hp, c.d_state, index, g_ctx.opt_pingpong ? (index & 1) : 0));   // :722
        if (box) {
            launch_cg_step<0>(a, s);
        } else {
            launch_cg_step<1>(a, s);
            BH_TRY(launch_project(P, c.r, c.v, c.d_state));            // :741
            launch_cg_step<2>(a, s);
        }
        return BH_OK;
    };

    // Launch schedule.  Kernels of iterations past the exit see state->done and return at once, so over-launching is
    // cheap (~3 trivial kernels per iteration) but not free.  First a batch sized by the previous call on this handle
    // (consecutive subproblems of a minor loop behave alike), then, if the loop is still running, launch-ahead batches:
    // batch k+1 is enqueued before the host looks at batch k's state, so the GPU never waits for the host.
    const int batch = launch_batch_size(H);
    int launched = 0;
    auto launch_batch = [&](int nb) -> int32_t {
        nb = std::min(nb, max_iter - launched);
        for (int i = 0; i < nb; ++i) BH_TRY(launch_iteration(launched + i));
        launched += nb;
        return BH_OK;
    };
    // Progress comes back through one host-mapped 8-byte word the kernels store to (no copy kernels, no events).
    // Multi-rank lock-step: every rank must take the SAME launch decisions (each launched iteration contains an
    // all-reduce), so a decision taken after waiting for `target` iterations may only use "the loop had exited by
    // iteration `target`" — never a later state that a slower-polling rank happened to see (the word keeps advancing
    // while launch-ahead batches run).  done_by(target) is that rank-independent predicate.
    MirrorWord mw{};
    auto done_by = [&](int target) { return mw.done && mw.n_hmul <= target; };
    // (an RCCL all-reduce cannot be gated from the device: every over-launched iteration pays for one.  With history on the handle
    // the first batch is the previous call's count exactly — the prediction the two-kernel form uses for its stop_at launch — so a
    // repeated subproblem enqueues as many collectives as it has H*p products; without history it stays small.)
    const bool rccl_path = comm_active() && !use_peer_path();
    const int first = H->last_n_hmul > 0 ? std::min(H->last_n_hmul, kFirstBatchCap) : std::min(batch, rccl_path ? 1 : 2);
    BH_TRY(launch_batch(first));
    BH_TRY(wait_mirror(c, a.tag, launched, &mw));
    if (!done_by(launched) && launched < max_iter) {
        BH_TRY(launch_batch(batch));
        while (true) {
            const int target = launched;             // everything enqueued so far except the batch launched next
            const bool more = launched < max_iter;
            if (more) BH_TRY(launch_batch(batch));
            BH_TRY(wait_mirror(c, a.tag, target, &mw));
            if (done_by(target) || !more) break;
        }
    }
    BH_TRY(wait_mirror(c, a.tag, launched, &mw));      // the final state (all enqueued iterations have run or were no-ops)
    if (!mw.done) return fail(BH_ERR_HIP, "internal: CG loop did not terminate");
    fin_out->done = mw.done; fin_out->status = mw.status; fin_out->iter = mw.iter; fin_out->n_hmul = mw.n_hmul;
    fin_out->tag = a.tag;
    return BH_OK;
}

// The tie log the final publish_state left next to the progress word.  The words are stored before the progress word by the
// same thread, but only a drained stream makes that an ordering guarantee.
static void read_tie_words(bh_hess* H) {
    const unsigned long long tw = g_ctx.cg.h_mirror[1], mb = g_ctx.cg.h_mirror[2];
    H->margin_kind = (int)((tw >> 48) & 0xf); H->tie_flags = (int)((tw >> 40) & 0xff);
    H->tie_first = (int)((tw >> 20) & 0xfffff); H->margin_at = (int)(tw & 0xfffff);
    memcpy(&H->min_margin, &mb, sizeof(double));
    H->tie_read = true;
}

// Bookkeeping of one finished projected_cg (normally after the caller's hipStreamSynchronize; `drained` = false when the
// device-pointer entry point returned on results_final without draining).
static int32_t pcg_finish(bh_hess* H, const PcgFin& fin, bool drained = true) {
    if (!fin.done) return fail(BH_ERR_HIP, "internal: CG loop did not terminate");
    H->stats.n_pcg += 1;
    H->last_n_hmul = fin.n_hmul;
    H->tie_tag = fin.tag;
    H->tie_read = false;
    if (drained) read_tie_words(H);     // else: read on demand (bh_pcg_tie_info drains first)
    H->stats.n_hmul += fin.n_hmul;
    H->stats.n_cg_iter += fin.iter - 1;
    H->stats.n_proj += fin.iter;
    if (!H->ev_pending.empty()) {
        for (size_t i = 0; i < H->ev_pending.size(); ++i) {
            if (H->ev_pending[i] >= fin.n_hmul) continue;         // an over-launched no-op: not an H*p
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, H->ev[2 * i], H->ev[2 * i + 1]) == hipSuccess) {
                H->stats.hmul_ms += ms;
                H->stats.hmul_timed += 1;
            }
        }
        H->ev_pending.clear();
    }
    return BH_OK;
}

static int32_t pcg_impl(bh_hess* H, bh_proj* P, const double* g_minor, const double* w_l, const double* w_u, double kappa2,
                        double atol_negcurv, double atol_f2b, double* w_out, int32_t* status, int32_t* iters, double* trace,
                        int64_t trace_cap, int32_t* n_hmul_out, bool dev) {
    BH_REQUIRE_INIT();
    if (!H) return fail(BH_ERR_INVALID_ARG, "NULL bh_hess");
    BH_TRY(check_proj_ready(P));
    if (!g_minor || !w_l || !w_u || !w_out) return fail(BH_ERR_INVALID_ARG, "NULL vector argument");
    if (H->n != P->n) return fail(BH_ERR_SHAPE, "H.n != lincons.n");
    if (trace == nullptr) trace_cap = 0;
    if (trace_cap < 0) return fail(BH_ERR_INVALID_ARG, "negative trace_cap");
    const int64_t n = H->n;
    BH_TRY(ensure_cg_workspace(H->ld, trace_cap));
    CgWorkspace& c = g_ctx.cg;
    // Device callers with even n and 16-byte aligned buffers hand over memory the kernels can use in place (16-byte chunk
    // loads stay in bounds); host callers, odd n and unaligned pointers go through the zero-padded workspace.
    auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    // With a communicator the choice must not depend on this rank's pointers: in place only when n == ld, where the CG iteration
    // shape chosen in pcg_run (two-kernel / three-kernel, which exchange different payloads and round pHp differently) is the
    // same whether a rank works in place or through the workspace; below ld every rank stages.
    const bool in_place = dev && (n % 2 == 0) && aligned16(g_minor) && aligned16(w_l) && aligned16(w_u) && aligned16(w_out) &&
                          (!comm_active() || n == H->ld);
    const double *gp = g_minor, *wlp = w_l, *wup = w_u;
    double* wp = w_out;
    if (!in_place) {
        if (!dev) {
            BH_TRY(stage_vecs(c.g, {g_minor, w_l, w_u}, n, c.n_pad));      // c.g, c.wl, c.wu are consecutive
        } else {
            BH_TRY(stage_vec(c.g, g_minor, n, true));
            BH_TRY(stage_vec(c.wl, w_l, n, true));
            BH_TRY(stage_vec(c.wu, w_u, n, true));
        }
        gp = c.g; wlp = c.wl; wup = c.wu; wp = c.w;
    }
    PcgFin fin{};
    BH_TRY(pcg_run(H, P, gp, wlp, wup, wp, !in_place, kappa2, atol_negcurv, atol_f2b, trace_cap, &fin, nullptr, !dev));
    if (!in_place) BH_TRY(fetch_vec(w_out, c.w, n, dev));
    if (trace_cap > 0) count_d2h((size_t)4 * trace_cap * sizeof(double));
    if (trace_cap > 0) note_dma();
    if (trace_cap > 0) BH_HIP(hipMemcpyAsync(trace, c.d_trace, (size_t)4 * trace_cap * sizeof(double), hipMemcpyDeviceToHost, g_ctx.stream));
    // The final hipStreamSynchronize costs ~10 us (measured: 650 -> 640 us per config-3 subproblem).  It is skipped when nothing
    // can still be written: device vectors used in place (no staged result to fetch), no trace, and the loop stopped in the
    // prologue of an H*p launch (PcgFin::results_final) — w was complete before that launch began.  Otherwise the stream is
    // drained as before (option "final_sync" = 1 forces it).
    const bool drain = g_ctx.opt_final_sync || !(in_place && trace_cap == 0 && fin.results_final);
    if (drain) BH_TRY(sync_flush());      // drains the over-launched no-op kernels; orders w for any consumer
    BH_TRY(pcg_finish(H, fin, drain));
    if (status) *status = fin.status;
    if (iters) *iters = fin.iter;
    if (n_hmul_out) *n_hmul_out = fin.n_hmul;
    return BH_OK;
}

int32_t bh_pcg(bh_hess* H, bh_proj* P, const double* g_minor, const double* w_l, const double* w_u, double kappa2,
               double atol_negcurv, double atol_f2b, double* w_out, int32_t* status, int32_t* iters, double* trace,
               int64_t trace_cap, int32_t* n_hmul) {
    return pcg_impl(H, P, g_minor, w_l, w_u, kappa2, atol_negcurv, atol_f2b, w_out, status, iters, trace, trace_cap, n_hmul, false);
}

int32_t bh_pcg_tie_info(const bh_hess* H, int32_t* tie_flags, int32_t* first_tie_hmul, double* min_margin, int32_t* min_margin_kind,
                        int32_t* min_margin_hmul) {
    if (!H) return fail(BH_ERR_INVALID_ARG, "NULL bh_hess");
    if (!H->tie_read) {
        // the call returned without draining the stream: do it now, and make sure no later call has overwritten the words
        bh_hess* Hm = const_cast<bh_hess*>(H);
        if (g_ctx.init && g_ctx.stream) BH_HIP(hipStreamSynchronize(g_ctx.stream));
        if ((unsigned)((g_ctx.cg.h_mirror[0] >> 48) & 0xffffu) != (H->tie_tag & 0xffffu))
            return fail(BH_ERR_PRECONDITION, "bh_pcg_tie_info: another projected_cg has run since this handle's last one (ask right after the call)");
        read_tie_words(Hm);
    }
    if (tie_flags) *tie_flags = H->tie_flags;
    if (first_tie_hmul) *first_tie_hmul = H->tie_first;
    if (min_margin) *min_margin = H->min_margin;
    if (min_margin_kind) *min_margin_kind = H->margin_kind;
    if (min_margin_hmul) *min_margin_hmul = H->margin_at;
    return BH_OK;
}

int32_t bh_pcg_dev(bh_hess* H, bh_proj* P, const double* g_minor_dev, const double* w_l_dev, const double* w_u_dev, double kappa2,
                   double atol_negcurv, double atol_f2b, double* w_out_dev, int32_t* status, int32_t* iters, double* trace,
                   int64_t trace_cap, int32_t* n_hmul) {
    return pcg_impl(H, P, g_minor_dev, w_l_dev, w_u_dev, kappa2, atol_negcurv, atol_f2b, w_out_dev, status, iters, trace, trace_cap,
                    n_hmul, true);
}


// linesearch(g_model, H, w, w_l, w_u, fix_bounds) — src/basic_tralcnlss.jl:766-791 (device part shared with bh_minor_iterate):
// wHw = vthv(H, w_pad) -> H->scalar (all-reduced), then alpha -> c.scalars[0]; optionally w_pad *= alpha.
// hw: H*w accumulated by the CG loop (then w'Hw = w.hw, no sweep over J), or NULL (then wHw = vthv(H, w): one J*v pass).
static int32_t launch_linesearch(bh_hess* H, bh_proj* P, const double* g_dev, double* w_pad, const double* wl_dev, const double* wu_dev,
                                 bool scale_w, double* hw = nullptr, double* alpha_dst = nullptr /* default: c.scalars[0] */) {
    CgWorkspace& c = g_ctx.cg;
    if (hw == nullptr) {
        BH_TRY(launch_jv(H, w_pad, nullptr, true, H->scalar));
        BH_TRY(allreduce_inplace(H->scalar, 1, H));
        H->stats.n_jv += 1;
    }
    hipLaunchKernelGGL(linesearch_kernel, dim3(1), dim3(CG_T), 0, g_ctx.stream, g_dev, w_pad, wl_dev, wu_dev,
                       P->nfix > 0 ? P->fixrank : (const int*)nullptr, (const double*)H->scalar, hw, (int)H->n, scale_w ? 1 : 0,
                       alpha_dst ? alpha_dst : c.scalars);
    BH_HIP(hipGetLastError());
    return BH_OK;
}

static int32_t linesearch_impl(bh_hess* H, bh_proj* P, const double* g_model, const double* w, const double* w_l, const double* w_u,
                               double* alpha_out, bool dev) {
    BH_REQUIRE_INIT();
    if (!H) return fail(BH_ERR_INVALID_ARG, "NULL bh_hess");
    BH_TRY(check_proj_ready(P));
    if (!g_model || !w || !w_l || !w_u || !alpha_out) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    if (H->n != P->n) return fail(BH_ERR_SHAPE, "H.n != lincons.n");
    const int64_t n = H->n;
    BH_TRY(ensure_cg_workspace(H->ld, 0));
    CgWorkspace& c = g_ctx.cg;
    if (dev) {
        // the line-search kernel reads its vectors element by element (used where they lie); only vthv(H, w) needs w in whole
        // 16-byte chunks up to the padded length
        const double* wp = c.w;
        if (n == H->ld && (reinterpret_cast<uintptr_t>(w) & 15u) == 0) wp = w;
        else {
            BH_HIP(hipMemsetAsync(c.w, 0, (size_t)H->ld * sizeof(double), g_ctx.stream));
            BH_TRY(stage_vec(c.w, w, n, true));
        }
        BH_TRY(mbox_ensure());
        BH_TRY(launch_linesearch(H, P, g_model, const_cast<double*>(wp), w_l, w_u, false, nullptr, mbox_dev<double>(kMbScal)));
        BH_TRY(mbox_seal_and_wait(sizeof(double)));
        *alpha_out = *mbox_host<double>(kMbScal);
        return BH_OK;
    }
    BH_HIP(hipMemsetAsync(c.w, 0, (size_t)H->ld * sizeof(double), g_ctx.stream));
    BH_TRY(stage_vec(c.w, w, n, false));
    BH_TRY(stage_vec(c.g, g_model, n, false));
    BH_TRY(stage_vec(c.wl, w_l, n, false));
    BH_TRY(stage_vec(c.wu, w_u, n, false));
    BH_TRY(launch_linesearch(H, P, c.g, c.w, c.wl, c.wu, false));
    BH_TRY(fetch_vec(alpha_out, c.scalars, 1, false));
    BH_TRY(sync_flush());
    return BH_OK;
}
int32_t bh_linesearch(bh_hess* H, bh_proj* P, const double* g_model, const double* w, const double* w_l, const double* w_u,
                      double* alpha_out) {
    return linesearch_impl(H, P, g_model, w, w_l, w_u, alpha_out, false);
}
int32_t bh_linesearch_dev(bh_hess* H, bh_proj* P, const double* g_model_dev, const double* w_dev, const double* w_l_dev,
                          const double* w_u_dev, double* alpha_out) {
    return linesearch_impl(H, P, g_model_dev, w_dev, w_l_dev, w_u_dev, alpha_out, true);
}

// minor_iterate(x, s, g_model, H, lincons, delta, kappa2) — src/basic_tralcnlss.jl:649-675, entirely on the device:
// step bounds (:660-665), projected_cg (:667), linesearch + scaling (:669-672).
static int32_t minor_iterate_impl(bh_hess* H, bh_proj* P, const double* x, const double* s_vec, const double* g_model, const double* xlow,
                                  const double* xupp, double delta, double kappa2, double atol_negcurv, double atol_f2b, double* w_out,
                                  int32_t* status, int32_t* iters, int32_t* n_hmul_out, double* alpha_out, bool dev) {
    BH_REQUIRE_INIT();
    if (!H) return fail(BH_ERR_INVALID_ARG, "NULL bh_hess");
    BH_TRY(check_proj_ready(P));
    if (!x || !s_vec || !g_model || !xlow || !xupp || !w_out) return fail(BH_ERR_INVALID_ARG, "NULL vector argument");
    if (H->n != P->n) return fail(BH_ERR_SHAPE, "H.n != lincons.n");
    const int64_t n = H->n;
    BH_TRY(ensure_cg_workspace(H->ld, 0));
    CgWorkspace& c = g_ctx.cg;
    // device callers: x, s and the bounds are only read element-wise (no staging); g_minor is copied device-to-device into
    // the zero-padded workspace vector the CG kernels read in 16-byte chunks
    const double *xp = x, *sp = s_vec, *lop = xlow, *upp = xupp;
    const double* gp = c.g;
    double* wp = c.w;
    // ls_from_cg: the CG loop accumulates H*w next to w, so linesearch's w'Hw (vthv(H,w), :775) costs a dot product
    // instead of another sweep over J (-0.3 ms per minor iterate at config 3); mathematically identical, rounding ~1e-15.
    double* hw = g_ctx.opt_ls_from_cg ? c.hw : nullptr;
    if (!dev) {
        BH_TRY(stage_vecs(c.s, {s_vec, x, xlow, xupp, g_model}, n, c.n_pad));      // c.s, c.x, c.xlow, c.xupp, c.g are consecutive
        xp = c.x; sp = c.s; lop = c.xlow; upp = c.xupp;
        BH_TRY(mbox_ensure());
    } else {
        // device callers: x, s and the bounds are only read element-wise; g_minor and w are used where they lie when the
        // kernels' 16-byte chunk accesses stay inside them (else through the zero-padded workspace)
        BH_TRY(device_operand(&gp, c.g, g_model, n, H->ld));
        // (w: the two-kernel iteration reads and writes whole 16-byte chunks up to the padded length)
        const bool w_ok = (n == H->ld) && (reinterpret_cast<uintptr_t>(w_out) & 15u) == 0;
        if (w_ok) wp = w_out;
        BH_TRY(mbox_ensure());
    }
    const int grid = std::max(1, std::min((int)((n + 255) / 256), 1024));
    hipLaunchKernelGGL(step_bounds_kernel, dim3(grid), dim3(256), 0, g_ctx.stream, xp, sp, lop, upp,
                       P->nfix > 0 ? P->fixrank : (const int*)nullptr, delta, (int)n, c.wl, c.wu);
    PcgFin fin{};
    BH_TRY(pcg_run(H, P, gp, c.wl, c.wu, wp, wp == c.w, kappa2, atol_negcurv, atol_f2b, 0, &fin, hw, !dev));
    double alpha = std::nan("");
    const bool do_ls = fin.status != BH_CG_NEGATIVE_CURVATURE;       // :669
    // alpha is written by the line search straight into the host-mapped mailbox (no DMA of its own)
    if (do_ls) BH_TRY(launch_linesearch(H, P, gp, wp, c.wl, c.wu, true, hw, mbox_dev<double>(kMbScal)));
    if (wp == c.w) BH_TRY(fetch_vec(w_out, c.w, n, dev));
    if (!dev) {
        BH_TRY(sync_flush());                               // w comes back through the pinned arena: the stream is drained
        if (do_ls) { count_d2h(sizeof(double)); alpha = *mbox_host<double>(kMbScal); }
    } else if (do_ls) {
        BH_TRY(mbox_seal_and_wait(sizeof(double)));        // alpha: written by the line search straight into the mailbox
        alpha = *mbox_host<double>(kMbScal);
    } else {
        BH_TRY(finish_device_call());
    }
    BH_TRY(pcg_finish(H, fin, !dev));
    // cg.hw now holds H*w for the w just delivered (scaled with it by the line search): bh_step_accumulate_dev may use it
    if (dev && hw != nullptr) { g_ctx.hw_note.H = H; g_ctx.hw_note.w = w_out; g_ctx.hw_note.gm = g_model; }
    if (status) *status = fin.status;
    if (iters) *iters = fin.iter;
    if (n_hmul_out) *n_hmul_out = fin.n_hmul;
    if (alpha_out) *alpha_out = alpha;
    return BH_OK;
}

int32_t bh_minor_iterate(bh_hess* H, bh_proj* P, const double* x, const double* s_vec, const double* g_model, const double* xlow,
                         const double* xupp, double delta, double kappa2, double atol_negcurv, double atol_f2b, double* w_out,
                         int32_t* status, int32_t* iters, int32_t* n_hmul_out, double* alpha_out) {
    return minor_iterate_impl(H, P, x, s_vec, g_model, xlow, xupp, delta, kappa2, atol_negcurv, atol_f2b, w_out, status, iters, n_hmul_out,
                              alpha_out, false);
}
int32_t bh_minor_iterate_dev(bh_hess* H, bh_proj* P, const double* x_dev, const double* s_dev, const double* g_model_dev,
                             const double* xlow_dev, const double* xupp_dev, double delta, double kappa2, double atol_negcurv,
                             double atol_f2b, double* w_out_dev, int32_t* status, int32_t* iters, int32_t* n_hmul_out, double* alpha_out) {
    return minor_iterate_impl(H, P, x_dev, s_dev, g_model_dev, xlow_dev, xupp_dev, delta, kappa2, atol_negcurv, atol_f2b, w_out_dev, status,
                              iters, n_hmul_out, alpha_out, true);
}

// g = Jx'*rx + Cx'*y_bar — src/basic_tralcnlss.jl:45,:74 (r = this rank's d rows; C'y_bar added by rank 0; all-reduced).
static int32_t adopt_device_mask(bh_proj* P, uint64_t* fix_chunks_out, int* info_out, int* counts_out = nullptr);

static int32_t grad_impl(bh_hess* H, const double* r, const double* ybar, double* g_out, bool dev) {
    BH_REQUIRE_INIT();
    if (!H || (!r && H->d > 0) || (!ybar && H->q > 0) || !g_out) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    BH_TRY(stage_vec(H->upad, r, H->d, dev));
    BH_TRY(stage_vec(H->upad + H->d, ybar, H->q, false));         // q multipliers: always a (tiny) host vector
    BH_TRY(launch_jtv(H, H->upad, H->zpad, true));
    BH_TRY(fetch_vec(g_out, H->zpad, H->n, dev));
    BH_TRY(sync_flush());          // (ybar is a host vector in both forms: the pinned arena has traffic in flight)
    H->stats.n_jtv += 1;
    return BH_OK;
}
int32_t bh_grad(bh_hess* H, const double* r, const double* ybar, double* g_out) { return grad_impl(H, r, ybar, g_out, false); }
int32_t bh_grad_dev(bh_hess* H, const double* r_dev, const double* ybar, double* g_out_dev) { return grad_impl(H, r_dev, ybar, g_out_dev, true); }

// dot(rx, rx) of mx = 0.5*dot(rx,rx) + ... — src/basic_tralcnlss.jl:44 (new_point), :58 (evaluate_al): r = this rank's rows of
// the residual; the partial sums of squares are all-reduced (rank order), so every rank gets the same bits.
int32_t bh_resid_sqnorm(const double* r, int64_t d, double* out) {
    BH_REQUIRE_INIT();
    if (d < 0 || (!r && d > 0) || !out) return fail(BH_ERR_INVALID_ARG, "bad argument");
    if (d + 4 > g_ctx.rbuf_cap) {
        dev_free(g_ctx.rbuf);
        g_ctx.rbuf = nullptr; g_ctx.rbuf_cap = 0;
        BH_TRY(dev_alloc(&g_ctx.rbuf, d + 4));
        g_ctx.rbuf_cap = d + 4;
    }
    double* acc = g_ctx.rbuf + round_up(d, 2);   // 2 doubles, 16-byte aligned: the exchange works on whole chunks
    BH_TRY(stage_vec(g_ctx.rbuf, r, d, false));
    hipLaunchKernelGGL(weighted_sqsum_kernel, dim3(1), dim3(1024), 0, g_ctx.stream, (const double*)g_ctx.rbuf, d, d, 1.0, acc);
    BH_HIP(hipGetLastError());
    BH_TRY(allreduce_inplace(acc, 1, nullptr));
    BH_TRY(fetch_vec(out, acc, 1, false));
    BH_TRY(sync_flush());
    return BH_OK;
}

// g_minor = H*s + g — src/basic_tralcnlss.jl:412,:437.
static int32_t hmul_add_impl(bh_hess* H, const double* s_vec, const double* g, double* out_n, bool dev, const double* w_add = nullptr,
                             double* s_inout = nullptr) {
    BH_REQUIRE_INIT();
    if (!H || !s_vec || !g || !out_n) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    const int64_t n = H->n;
    BH_TRY(ensure_cg_workspace(H->ld, 0));
    const int grid = std::max(1, std::min((int)((n + 255) / 256), 1024));
    if (w_add != nullptr)      // s .+= w (:436) on the caller's device vector, then H*s + g
        hipLaunchKernelGGL(vec_add_kernel, dim3(grid), dim3(256), 0, g_ctx.stream, (const double*)s_inout, w_add, s_inout, (int)n);
    const double* sp = H->vpad;
    if (dev) BH_TRY(device_operand(&sp, H->vpad, s_vec, n, H->ld));
    else BH_TRY(stage_vec(H->vpad, s_vec, n, false));
    BH_TRY(launch_hmul(H, sp, H->zpad, nullptr, -1));
    if (dev) {
        hipLaunchKernelGGL(vec_add_kernel, dim3(grid), dim3(256), 0, g_ctx.stream, (const double*)H->zpad, g, out_n, (int)n);
        BH_HIP(hipGetLastError());
    } else {
        CgWorkspace& c = g_ctx.cg;
        BH_TRY(stage_vec(c.g, g, n, false));
        hipLaunchKernelGGL(vec_add_kernel, dim3(grid), dim3(256), 0, g_ctx.stream, (const double*)H->zpad, (const double*)c.g, H->zpad, (int)n);
        BH_TRY(fetch_vec(out_n, H->zpad, n, false));
    }
    BH_TRY(dev ? finish_device_call() : sync_flush());
    H->stats.n_hmul += 1;
    if (dev) g_ctx.gm_note = {H, s_vec, g, out_n}; else g_ctx.gm_note = {};
    return BH_OK;
}
int32_t bh_hmul_add(bh_hess* H, const double* s_vec, const double* g, double* out_n) { return hmul_add_impl(H, s_vec, g, out_n, false); }
int32_t bh_hmul_add_dev(bh_hess* H, const double* s_dev, const double* g_dev, double* out_n_dev) {
    return hmul_add_impl(H, s_dev, g_dev, out_n_dev, true);
}
// s .+= w ; g_minor = H*s + g — src/basic_tralcnlss.jl:436-437, on device vectors (s is updated in place).
int32_t bh_step_accumulate_dev(bh_hess* H, double* s_dev, const double* w_dev, const double* g_dev, double* g_minor_out_dev) {
    if (!w_dev || !s_dev) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    // The inner step's pattern (src/basic_tralcnlss.jl:434-437): g_minor_out holds H*s + g, it was the g_model of the
    // bh_minor_iterate_dev that has just produced w, and that call's CG loop accumulated H*w.  Then H*(s + w) + g = g_minor + H*w:
    // two n-vector kernels instead of a sweep over J.  Recognised by the three pointers; anything else recomputes H*s + g.
    if (g_ctx.init && H && g_dev && g_minor_out_dev && g_ctx.opt_step_from_cg && g_ctx.hw_note.H == H && g_ctx.hw_note.w == w_dev &&
        g_ctx.hw_note.gm == g_minor_out_dev && g_ctx.cg.hw != nullptr) {
        g_ctx.hw_note = {};                          // one use: s changes below, a second call with the same w is a different step
        const int64_t n = H->n;
        const int grid = std::max(1, std::min((int)((n + 255) / 256), 1024));
        hipLaunchKernelGGL(vec_add_kernel, dim3(grid), dim3(256), 0, g_ctx.stream, (const double*)s_dev, w_dev, s_dev, (int)n);
        hipLaunchKernelGGL(vec_add_kernel, dim3(grid), dim3(256), 0, g_ctx.stream, (const double*)g_minor_out_dev, (const double*)g_ctx.cg.hw,
                           g_minor_out_dev, (int)n);
        BH_HIP(hipGetLastError());
        g_ctx.gm_note = {H, s_dev, g_dev, g_minor_out_dev};
        return finish_device_call();
    }
    g_ctx.hw_note = {};
    return hmul_add_impl(H, s_dev, g_dev, g_minor_out_dev, true, w_dev, s_dev);
}

// active_indx = active_bounds(lincons, x, s, delta); add_active!(lincons, chol_aat, active_indx) — or, when mA + |active_indx| > n,
// active_bounds!(lincons, x+s, chol_aat) — src/basic_tralcnlss.jl:439-453, src/polyhedral_constraints.jl:203-261, on the device:
// the active set lives in HBM, newly fixed variables are taken out of A_free A_free' by a Gram DOWNDATE over just their
// columns (then the mA x mA refactorisation) instead of the reference's from-scratch O(p^3) cholesky_aug_aat rebuild; only
// three counts come back.
int32_t bh_proj_update_active_dev(bh_proj* P, const double* x_dev, const double* s_dev, const double* xlow_dev, const double* xupp_dev,
                                  double delta, double atol, int32_t* n_at_bound, int32_t* n_fixed, int32_t* branch, uint64_t* fix_chunks_out) {
    BH_REQUIRE_INIT();
    if (!P || !x_dev || !s_dev || !xlow_dev || !xupp_dev) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    if (P->mA > 0 && g_ctx.opt_proj_form == 0)
        return fail(BH_ERR_UNSUPPORTED, "bh_proj_update_active_dev needs the reduced projection form (proj_form = 1)");
    const int mA = (int)P->mA;
    hipStream_t s = g_ctx.stream;
    if (mA > 0) {
        BH_TRY(ensure_reduced_buffers(P));
        const size_t lds = trsv_lds_bytes(mA);
        if (lds > kLdsPerCu) return fail(BH_ERR_UNSUPPORTED, "factor too large for the single-workgroup triangular solve");
        BH_TRY(ensure_trsv_lds(lds));
    }
    const bool had_factor = P->active_set && P->reduced && P->M_valid;   // M and Lr describe the CURRENT active set: a downdate is enough
    hipLaunchKernelGGL(active_update_kernel, dim3(1), dim3(CG_T), 0, s, x_dev, s_dev, xlow_dev, xupp_dev, delta, atol, (int)P->n, (int)P->ldA,
                       mA, P->fixrank, P->newidx, P->counts);
    BH_HIP(hipGetLastError());
    int counts[4] = {0, 0, 0, 0};
    if (mA > 0) {
        // with linear equalities the host chooses between a Gram downdate and a rebuild: it needs the counts now (mailbox round
        // trip); box constraints need nothing between the two kernels, their counts come back with the mask below
        BH_TRY(mbox_ensure());
        BH_TRY(mbox_seal_and_wait(sizeof(counts), nullptr, nullptr, nullptr, P->counts, 4, mbox_dev<int>(kMbInts)));
        for (int i = 0; i < 4; ++i) counts[i] = mbox_host<int>(kMbInts)[i];
        if (counts[AU_BRANCH] == 0 && had_factor) {
            if (counts[AU_NEW] > 0) {
                hipLaunchKernelGGL(gram_downdate_list_kernel, dim3(std::max(1, std::min(1024, (mA * mA + 255) / 256))), dim3(256), 0, s, P->M,
                                   (const double*)P->Ad, P->ldA, mA, (const int*)P->newidx, (const int*)P->counts);
                BH_TRY(launch_chol(P, nullptr));
            }
        } else {
            BH_TRY(launch_reduced_factor(P, true, nullptr));      // active set replaced (or no factor yet): from the mask
        }
    }
    P->active_set = false;
    int info_host = 0;
    int counts_now[4] = {0, 0, 0, 0};
    BH_TRY(adopt_device_mask(P, fix_chunks_out, &info_host, counts_now));   // (canon_mask_kernel leaves AU_AT_BOUND / AU_BRANCH untouched)
    if (info_host != 0) {
        P->active_set = false;
        return fail(BH_ERR_PRECONDITION, "A_free*A_free' is not positive definite (PosDefException in the reference's cholesky)");
    }
    if (mA == 0) for (int i = 0; i < 4; ++i) counts[i] = counts_now[i];                  // arrived with the mask
    if (n_at_bound) *n_at_bound = counts[AU_AT_BOUND];
    if (n_fixed) *n_fixed = P->nfix;
    if (branch) *branch = counts[AU_BRANCH];
    return BH_OK;
}

// norm_reduced_gradient(g, lincons) = norm(projection(lincons, -g)) — src/basic_tralcnlss.jl:869-875 (criticality_measure :839).
// The projector is linear and every operation in it is sign-symmetric in IEEE arithmetic, so ||P(-g)|| and ||P(g)|| are the
// same bits: g is projected as it is.
int32_t bh_reduced_gradient_norm_dev(bh_proj* P, const double* g_dev, double* out) {
    BH_REQUIRE_INIT();
    BH_TRY(check_proj_ready(P));
    if (!g_dev || !out) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    BH_TRY(mbox_ensure());
    double* res = mbox_dev<double>(kMbScal);
    if (P->mA == 0) {
        // box constraints: projection = mask, fused with the norm (one launch, the caller's vector read in place)
        hipLaunchKernelGGL(vec_norm_masked_kernel, dim3(1), dim3(CG_T), 0, g_ctx.stream, g_dev, P->nfix > 0 ? (const int*)P->fixrank : (const int*)nullptr,
                           (int)P->n, res);
    } else {
        // a vector that needs no padding (n a multiple of 16, 16-byte aligned) is projected where it lies
        const double* rp = g_dev;
        if (P->n != P->ldA || (reinterpret_cast<uintptr_t>(g_dev) & 15u) != 0) {
            BH_TRY(stage_vec(P->rpad, g_dev, P->n, true));
            rp = P->rpad;
        }
        BH_TRY(launch_project(P, rp, P->vtmp, nullptr));
        hipLaunchKernelGGL(vec_norm_kernel, dim3(1), dim3(CG_T), 0, g_ctx.stream, (const double*)P->vtmp, (int)P->n, res);
    }
    BH_HIP(hipGetLastError());
    BH_TRY(mbox_seal_and_wait(sizeof(double)));
    *out = *mbox_host<double>(kMbScal);
    return BH_OK;
}

// model_reduction = dot(g,s) + 0.5*vthv(H,s) — src/basic_tralcnlss.jl:458.
int32_t bh_model_reduction_dev(bh_hess* H, const double* g_dev, const double* s_dev, double* out) {
    BH_REQUIRE_INIT();
    if (!H || !g_dev || !s_dev || !out) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    BH_TRY(ensure_cg_workspace(H->ld, 0));
    CgWorkspace& c = g_ctx.cg;
    if (g_ctx.opt_step_from_cg && g_ctx.gm_note.H == H && g_ctx.gm_note.s == s_dev && g_ctx.gm_note.g == g_dev && g_ctx.gm_note.gm != nullptr) {
        // the library wrote g_minor = H*s + g for exactly these vectors (and, by the caller's opt-in, nobody has touched them since):
        // s'Hs = s.(g_minor - g), g.s — no sweep over J (the rounding of the difference is that of g.s itself: eps |g||s|)
        hipLaunchKernelGGL(model_from_gminor_kernel, dim3(1), dim3(CG_T), 0, g_ctx.stream, g_dev, s_dev, g_ctx.gm_note.gm, (int)H->n, c.scalars + 2);
        BH_HIP(hipGetLastError());
        BH_TRY(mbox_ensure());
        BH_TRY(mbox_seal_and_wait(2 * sizeof(double), c.scalars + 2, c.scalars + 3, mbox_dev<double>(kMbScal)));
        *out = mbox_host<double>(kMbScal)[1] + 0.5 * mbox_host<double>(kMbScal)[0];
        return BH_OK;
    }
    const double* sp = H->vpad;
    BH_TRY(device_operand(&sp, H->vpad, s_dev, H->n, H->ld));
    BH_TRY(launch_jv(H, sp, nullptr, true, H->scalar));
    BH_TRY(allreduce_inplace(H->scalar, 1, H));
    hipLaunchKernelGGL(vec_dot_kernel, dim3(1), dim3(CG_T), 0, g_ctx.stream, g_dev, s_dev, (int)H->n, c.scalars + 1);
    BH_HIP(hipGetLastError());
    BH_TRY(mbox_ensure());
    // s'Hs (all-reduced, so it lives in device memory) and g's ride to the host with the seal
    BH_TRY(mbox_seal_and_wait(2 * sizeof(double), H->scalar, c.scalars + 1, mbox_dev<double>(kMbScal)));
    const double vthv = mbox_host<double>(kMbScal)[0], gs = mbox_host<double>(kMbScal)[1];
    H->stats.n_jv += 1;
    *out = gs + 0.5 * vthv;
    return BH_OK;
}

// After the device-side active set (P->fixrank >= 0 <=> fixed) has changed — bh_cauchy_step, bh_proj_update_active_dev —
// bring the canonical device arrays (fixrank = rank among the fixed, fixidx) and the host bookkeeping in line with it
// WITHOUT moving the mask through the host: a scan kernel renumbers in place, and only the count, the BitVector image
// (n/8 bytes: what the caller's lincons.fixvars needs) and the factorisation flag come back.
static int32_t adopt_device_mask(bh_proj* P, uint64_t* fix_chunks_out, int* info_out, int* counts_out) {
    const int64_t n = P->n;
    const size_t nwords = (size_t)((n + 63) / 64);
    BH_TRY(mbox_ensure());
    const bool via_mbox = kMbChunks + nwords * sizeof(uint64_t) <= kMboxBytes - kMboxPayloadOff;
    std::vector<uint64_t> chunks(nwords, 0ull);
    int counts[5] = {0, 0, 0, 0, 0};          // AU_* and, for mA > 0, the factorisation flag (P->info = P->counts + 4)
    const bool with_info = P->mA > 0 && P->info != nullptr;
    if (via_mbox) {
        // the scan kernel writes the BitVector image straight into the mailbox; counts + flag ride with the seal
        hipLaunchKernelGGL(canon_mask_kernel, dim3(1), dim3(CG_T), 0, g_ctx.stream, P->fixrank, P->fixidx, (int)n, (int)P->ldA,
                           mbox_dev<unsigned long long>(kMbChunks), P->counts);
        BH_HIP(hipGetLastError());
        BH_TRY(mbox_seal_and_wait(nwords * sizeof(uint64_t) + sizeof(counts), nullptr, nullptr, nullptr, P->counts, with_info ? 5 : 4,
                                  mbox_dev<int>(kMbInts)));
        for (size_t i = 0; i < nwords; ++i) chunks[i] = mbox_host<unsigned long long>(kMbChunks)[i];
        for (int i = 0; i < (with_info ? 5 : 4); ++i) counts[i] = mbox_host<int>(kMbInts)[i];
    } else {
        hipLaunchKernelGGL(canon_mask_kernel, dim3(1), dim3(CG_T), 0, g_ctx.stream, P->fixrank, P->fixidx, (int)n, (int)P->ldA, P->chunks_dev, P->counts);
        BH_HIP(hipGetLastError());
        count_d2h(nwords * sizeof(uint64_t) + sizeof(counts));
        note_dma();
        BH_HIP(hipMemcpyAsync(chunks.data(), P->chunks_dev, nwords * sizeof(uint64_t), hipMemcpyDeviceToHost, g_ctx.stream));
        BH_HIP(hipMemcpyAsync(counts, P->counts, (with_info ? 5 : 4) * sizeof(int), hipMemcpyDeviceToHost, g_ctx.stream));
        BH_TRY(sync_flush());
    }
    P->nfix = counts[AU_FIXED]; P->mpp = (int)P->mA + P->nfix; P->reduced = P->mA > 0; P->have_L = false;
    P->last_chunks = chunks;
    P->active_set = true;
    if (fix_chunks_out) memcpy(fix_chunks_out, chunks.data(), nwords * sizeof(uint64_t));
    if (info_out) *info_out = with_info ? counts[4] : 0;
    if (counts_out) for (int i = 0; i < 4; ++i) counts_out[i] = counts[i];
    return BH_OK;
}

// cauchy_step(x, g, H, chol_aat, lincons, delta) — src/basic_tralcnlss.jl:574-639, device-resident: initial
// active_bounds! (poly:203-215), projection of -g, then per breakpoint one H*d, one projection and — instead of the
// reference's O(p^3) host refactorisation (add_active! -> cholesky_aug_aat) — a rank-one downdate of A_free A_free' and an
// mA x mA Cholesky on the device.  On return lincons' active set is the one the reference would hold (fix_chunks_out).
static int32_t cauchy_impl(bh_hess* H, bh_proj* P, const double* x, const double* g, const double* xlow, const double* xupp, double delta,
                           double* s_out, uint64_t* fix_chunks_out, int32_t* n_breakpoints, int32_t* n_hmul_out, bool dev) {
    BH_REQUIRE_INIT();
    if (!H || !P) return fail(BH_ERR_INVALID_ARG, "NULL handle");
    if (!x || !g || !xlow || !xupp || !s_out) return fail(BH_ERR_INVALID_ARG, "NULL vector argument");
    if (H->n != P->n) return fail(BH_ERR_SHAPE, "H.n != lincons.n");
    if (P->mA > 0 && g_ctx.opt_proj_form == 0)
        return fail(BH_ERR_UNSUPPORTED, "bh_cauchy_step needs the reduced projection form (proj_form = 1)");
    const int64_t n = H->n;
    if (n + 1 >= 0xfffff) return fail(BH_ERR_UNSUPPORTED, "n exceeds the 20-bit pass counter of the progress word");
    const int mA = (int)P->mA;
    BH_TRY(ensure_cg_workspace(H->ld, 0));
    CgWorkspace& c = g_ctx.cg;
    hipStream_t s = g_ctx.stream;
    if (mA > 0) {
        BH_TRY(ensure_reduced_buffers(P));
        const size_t lds = trsv_lds_bytes(mA);
        if (lds > kLdsPerCu) return fail(BH_ERR_UNSUPPORTED, "factor too large for the single-workgroup triangular solve");
        BH_TRY(ensure_trsv_lds(lds));
    }
    // (the Cauchy kernels read x, g and the bounds element by element: a device caller's vectors are used where they lie)
    if (!dev) BH_TRY(stage_vecs(c.x, {x, xlow, xupp, g}, n, c.n_pad));            // c.x, c.xlow, c.xupp, c.g are consecutive

    CauchyArgs a{};
    a.st = c.d_state;
    a.x = dev ? x : c.x; a.g = dev ? g : c.g; a.xlow = dev ? xlow : c.xlow; a.xupp = dev ? xupp : c.xupp;
    a.negg = c.r; a.d = c.p; a.Hd = c.Hp; a.s = c.w; a.dl = c.wl; a.du = c.wu;
    a.fixrank = P->fixrank; a.n = (int)n; a.n_pad = (int)H->ld; a.nmm = (int)(n - mA);
    a.delta = delta; a.atol = std::sqrt(2.220446049250313e-16);
    a.box = (mA == 0) ? 1 : 0;
    c.tag = (c.tag + 1) & 0xffffu;
    if (c.tag == 0) c.tag = 1;
    a.mirror = c.d_mirror; a.tag = c.tag;

    // Box constraints, one rank: the image-space search (bh_cauchy.hip.h) — t_d = J~ d once by the J v kernel, then per breakpoint
    // a rank-one update of t_d, t_s over the rows (one column of J) + the single-workgroup advance kernel; no sweep over J.
    // (several ranks: every rank keeps t_d, t_s for ITS rows; the two sums are all-reduced before the replicated advance kernel)
    // With linear equalities the form costs 1 + mA J v sweeps up front: always used up to cauchy_image_max_ma rows; up to 64 rows when
    // the previous search on this handle took more than 4 (1 + mA) passes (consecutive searches of a solve behave alike).
    const bool image = g_ctx.opt_cauchy_image != 0 &&
                       (mA == 0 || mA <= g_ctx.opt_cauchy_image_max_ma || (mA <= 64 && P->last_cauchy_passes > 4 * (1 + mA)));
    const bool image_gen = image && mA > 0;
    // ... and there ONE kernel per breakpoint: the decision of pass k-1 in the prologue of the row kernel of pass k (cauchy_fused_kernel)
    const bool fused = image && !image_gen && !comm_active() && g_ctx.opt_cauchy_fused != 0;
    const int64_t img_rows = H->d + H->q_eff;
    const int img_grid = (int)std::max<int64_t>(1, std::min<int64_t>(fused ? kCauchyFusedGrid : kCauchyImgGrid, (img_rows + 255) / 256));
    const int fused_grid = g_ctx.opt_cauchy_fused_grid > 0 ? (int)g_ctx.opt_cauchy_fused_grid
                                                           : (int)std::max<int64_t>(1, std::min<int64_t>(kCauchyFusedGrid, (img_rows + CA_T - 1) / CA_T));
    // with equalities a workgroup takes tiles of 64 rows; the partial sums still have to fit the [2][kCauchyImgGrid] slot
    const int gen_grid = (int)std::max<int64_t>(1, std::min<int64_t>(kCauchyImgGrid, (img_rows + 63) / 64));
    const bool gen_tiled = mA > 16;                                   // (few equalities: one row per thread, see bh_cauchy.hip.h)
    const int part_G = (image_gen && gen_tiled) ? gen_grid : img_grid;   // how many partial sums a row kernel leaves
    const int64_t img_cap = (std::max<int64_t>(H->d + H->q, 1) + 1) / 2 * 2;              // rows, rounded up to even (16-byte aligned tails)
    double* img_scal = nullptr;
    if (image) {
        BH_TRY(hess_ready(H));
        if (!H->timg) BH_TRY(dev_alloc(&H->timg, 2 * img_cap + 2 * kCauchyImgGrid + 16));
        img_scal = H->timg + 2 * img_cap + 2 * kCauchyImgGrid;
        a.img_part = comm_active() ? img_scal : H->timg + 2 * img_cap;
        a.img_G = comm_active() ? 1 : part_G;
        if (image_gen && H->timg_gen_doubles < (int64_t)(1 + mA) * img_cap) {
            dev_free(H->timg_gen);
            H->timg_gen = nullptr; H->timg_gen_doubles = 0;
            BH_TRY(dev_alloc(&H->timg_gen, (int64_t)(1 + mA) * img_cap));
            H->timg_gen_doubles = (int64_t)(1 + mA) * img_cap;
        }
    }

    CauchyPass* pp = nullptr;
    double* part_pp[2] = {nullptr, nullptr};
    double* sbuf[2] = {c.w, c.Hp};                                     // s_c ping-pong of the fused form (H*d is never formed there)
    if (fused) {
        pp = reinterpret_cast<CauchyPass*>(img_scal + 8);             // 2 x 32 bytes in the 16-double tail of timg
        part_pp[0] = H->timg + 2 * img_cap; part_pp[1] = part_pp[0] + 2 * kCauchyFusedGrid;
        a.fixpass = reinterpret_cast<int*>(c.v); a.pp0 = pp;
    }
    P->active_set = false;             // device mask is authoritative until adopt_mask below
    hipLaunchKernelGGL(cauchy_init_kernel, dim3(1), dim3(CG_T), 0, s, a);
    if (mA > 0) BH_TRY(launch_reduced_factor(P, true, nullptr));
    const int max_pass = (int)(n + 1);
    int launched = 0;
    auto launch_pass = [&](int index) -> int32_t {
        if (fused && index > 0) {
            CauchyFusedArgs fa{};
            fa.pp = pp; fa.k = index; fa.g = a.g; fa.dl = a.dl; fa.du = a.du; fa.sbuf[0] = sbuf[0]; fa.sbuf[1] = sbuf[1];
            fa.fixpass = a.fixpass; fa.fixrank = P->fixrank; fa.n = (int)n; fa.nmm = a.nmm;
            fa.J = H->Jd; fa.ld = H->ld; fa.nrows = img_rows; fa.d_rows = H->d; fa.mu = H->mu;
            fa.td = H->timg; fa.ts = H->timg + img_cap;
            fa.part_in = part_pp[(index - 1) & 1]; fa.Gin = index == 1 ? img_grid : fused_grid; fa.part_out = part_pp[index & 1];
            fa.mirror = a.mirror; fa.tag = a.tag;
            hipLaunchKernelGGL(cauchy_fused_kernel, dim3(fused_grid), dim3(CA_T), 0, s, fa);
            BH_HIP(hipGetLastError());
            return BH_OK;
        }
        if (fused) {                                                   // launch 0: t_d = J~ d_0 (:609 in the row space), t_s = 0, their sums
            CauchyImgArgs ia{};
            ia.st = c.d_state; ia.J = H->Jd; ia.ld = H->ld; ia.nrows = img_rows; ia.d_rows = H->d; ia.mu = H->mu;
            ia.td = H->timg; ia.ts = H->timg + img_cap; ia.part = part_pp[0]; ia.first = 1;
            BH_TRY(launch_jv(H, c.p, H->timg, true, nullptr));
            H->stats.n_jv += 1;
            hipLaunchKernelGGL(cauchy_image_kernel, dim3(img_grid), dim3(256), 0, s, ia);
            BH_HIP(hipGetLastError());
            return BH_OK;
        }
        if (image) {
            const int64_t rows_cap = img_cap;
            CauchyImgArgs ia{};
            ia.st = c.d_state; ia.J = H->Jd; ia.ld = H->ld; ia.nrows = img_rows; ia.d_rows = H->d; ia.mu = H->mu;
            ia.td = H->timg; ia.ts = H->timg + rows_cap; ia.part = H->timg + 2 * rows_cap; ia.first = index == 0 ? 1 : 0;
            if (image_gen) {
                // factor of the current active set and y = (A_free A_free')^{-1} A_free(-g) (left in P->tw), as in the sweeping form
                // Three launches per pass (image_gen implies mA <= 64): [downdate + refactorisation + right-hand side + the two solves: y]
                // -> [rows | d = P(-g) | t_fresh = A_free(-g) for the next pass] -> [decision]
                ProjArgs pa = proj_args(P, c.d_state, true, true);
                if (index > 0) {
                    P->linv_valid = false;
                    hipLaunchKernelGGL(cauchy_factor_solve_kernel, dim3(1), dim3(256), 0, s, P->M, P->Lr, mA, P->info, pa, (const double*)c.r,
                                       (const double*)P->tpart, (const CgState*)c.d_state);
                } else {
                    hipLaunchKernelGGL(proj_left_mul_kernel, dim3(mA), dim3(256), 0, s, pa, (const double*)c.r);     // t (:86-98), then y
                    hipLaunchKernelGGL(trsv_small_kernel, dim3(1), dim3(256), 0, s, pa);
                }
                BH_HIP(hipGetLastError());
                if (index == 0) {
                    // a = J~ D g: one J v sweep over the masked g.  B = J~ D A' (rows x mA): ONE sweep on the matrix cores
                    // (image_b_mfma_kernel) when the images share their leading dimension, else mA J v sweeps over masked rows of A
                    const int mgrid = std::max(1, std::min((int)((n + 255) / 256), 1024));
                    const bool gemm = g_ctx.opt_cauchy_gemm != 0 && P->ldA == H->ld;
                    for (int j = -1; j < (gemm ? 0 : mA); ++j) {
                        const double* src = (j < 0) ? a.g : (const double*)(P->Ad + (int64_t)j * P->ldA);
                        hipLaunchKernelGGL(proj_mask_kernel, dim3(mgrid), dim3(256), 0, s, src, H->vpad, (const int*)P->fixrank, (int)n, (const CgState*)nullptr);
                        BH_TRY(launch_jv(H, H->vpad, H->timg_gen + (int64_t)(j + 1) * rows_cap, true, nullptr));
                        H->stats.n_jv += 1;
                    }
                    if (gemm && img_rows > 0) {                 // (a rank without rows has no rows of B)
                        hipLaunchKernelGGL(image_b_mfma_kernel, dim3((unsigned)((img_rows + 127) / 128)), dim3(256), 0, s, (const double*)H->Jd, H->ld,
                                           img_rows, (const double*)P->Ad, P->ldA, mA, (const int*)P->fixrank, H->timg_gen + rows_cap, rows_cap);
                        BH_HIP(hipGetLastError());
                    }
                }
                CauchyImgGenArgs ga{};
                ga.b = ia; ga.a = H->timg_gen; ga.B = H->timg_gen + rows_cap; ga.rows_cap = rows_cap; ga.mA = mA;
                ga.A = P->Ad; ga.ldA = P->ldA; ga.tw = P->tw; ga.g = a.g;
                const int dblocks = ((int)(n + 1) / 2 + 63) / 64;
                hipLaunchKernelGGL(cauchy_gen_rows_and_d_kernel, dim3(part_G + dblocks + mA), dim3(256), 0, s, ga, gen_tiled ? 1 : 0, part_G, dblocks, pa,
                                   (const double*)c.r, c.p, P->tpart);
            } else {
                if (index == 0) {
                    BH_TRY(launch_jv(H, c.p, H->timg, true, nullptr));              // t_d = J~ d_0 (:609 in the row space)
                    H->stats.n_jv += 1;
                }
                hipLaunchKernelGGL(cauchy_image_kernel, dim3(img_grid), dim3(256), 0, s, ia);
            }
            if (comm_active()) {
                hipLaunchKernelGGL(cauchy_image_sum_kernel, dim3(1), dim3(64), 0, s, (const double*)ia.part, part_G, img_scal, (const CgState*)c.d_state);
                BH_TRY(allreduce_inplace(img_scal, 2, H, c.d_state));
            }
            hipLaunchKernelGGL(cauchy_advance_kernel, dim3(1), dim3(CA_T), 0, s, a);
            BH_HIP(hipGetLastError());
            return BH_OK;
        }
        if (index > 0 && mA > 0) {
            // chol_downdate = 1 only has a case where refactoring is expensive (mA > 64: blocked factorisation, 0.21 ms at mA = 256);
            // up to 64 rows the register-panel Cholesky (18.5 us) costs what the rank-one downdate costs (20 us), so the factor is
            // always rebuilt from the downdated Gram matrix there — as accurate as the reference's from-scratch rebuild.
            const bool factor_downdate = g_ctx.opt_chol_downdate && mA > 64;
            if (factor_downdate && (index % kDowndateRefresh) == 0) {
                // every kDowndateRefresh-th breakpoint the factor is rebuilt from the device-side mask: hyperbolic downdates lose
                // accuracy cumulatively, and without bound once A_free A_free' approaches singularity (c = sqrt(1 - s^2) -> 0)
                BH_TRY(launch_reduced_factor(P, true, (const CgState*)c.d_state));
            } else if (factor_downdate) {
                P->M_valid = false;     // only the factor follows the active set on this path
                // add_active!: one more fixed variable = rank-one downdate of chol(A_free A_free'), O(mA^2)
                P->linv_valid = false;
                hipLaunchKernelGGL(chol_downdate_kernel, dim3(1), dim3(CG_T), (size_t)mA * sizeof(double), s, P->Lr, (const double*)P->Ad,
                                   P->ldA, mA, P->info, (const CgState*)c.d_state);
            } else {
                // refactor from the downdated Gram matrix, O(mA^3)
                hipLaunchKernelGGL(gram_downdate_kernel, dim3(std::max(1, (mA * mA + 255) / 256)), dim3(256), 0, s, P->M, P->Ad, P->ldA, mA,
                                   (const CgState*)c.d_state);
                BH_TRY(launch_chol(P, (const CgState*)c.d_state));
            }
        }
        if (mA > 0) BH_TRY(launch_project(P, c.r, c.p, c.d_state, true));       // d = P(-g)   :592 / :632  (box: kept in place)
        BH_TRY(launch_hmul(H, c.p, c.Hp, c.d_state, -1));           // Hd = H*d    :609 / :633
        hipLaunchKernelGGL(cauchy_advance_kernel, dim3(1), dim3(CA_T), 0, s, a);
        BH_HIP(hipGetLastError());
        return BH_OK;
    };
    MirrorWord mw{};
    // image-space passes take ~17 us: keep a deeper queue ahead of the GPU (over RCCL every over-launched pass costs a collective)
    const int batch = !image ? launch_batch_size(H) : (comm_active() && !use_peer_path()) ? 4 : 8;
    // (fused form: launch k carries decision k-1, so `launched` launches stand for launched - 1 passes)
    const int off = fused ? 1 : 0;
    const int max_launch = max_pass + off;
    auto launch_batch = [&](int nb) -> int32_t {
        nb = std::min(nb, max_launch - launched);
        for (int i = 0; i < nb; ++i) BH_TRY(launch_pass(launched + i));
        launched += nb;
        return BH_OK;
    };
    // Same launch-ahead schedule and the same rank-independent exit predicate as pcg_run (mirror: status field = error
    // flag, iter field = breakpoints, n_hmul = passes run).
    auto done_by = [&](int target) { return mw.done && mw.n_hmul <= target; };
    BH_TRY(launch_batch(2 + off));
    while (true) {
        const int target = launched - off;
        const bool more = launched < max_launch;
        if (more) BH_TRY(launch_batch(batch));
        BH_TRY(wait_mirror(c, a.tag, target, &mw));
        if (done_by(target) || !more) break;
    }
    BH_TRY(wait_mirror(c, a.tag, launched - off, &mw));
    // (fused form: decision j is taken by launch j+1, which writes s_c into buffer (j+1) & 1)
    BH_TRY(fetch_vec(s_out, fused ? sbuf[mw.n_hmul & 1] : c.w, n, dev));
    int info_host = 0;
    BH_TRY(adopt_device_mask(P, fix_chunks_out, &info_host));     // drains the stream; canonical fixrank / fixidx, P->nfix
    if (!image) H->stats.n_hmul += mw.n_hmul;                  // (image-space search: passes, not sweeps over J)
    P->last_cauchy_passes = mw.n_hmul;
    if (!mw.done) return fail(BH_ERR_HIP, "internal: Cauchy loop did not terminate");
    if (n_breakpoints) *n_breakpoints = mw.iter;
    if (n_hmul_out) *n_hmul_out = mw.n_hmul;
    if (mw.status != 0)
        return fail(BH_ERR_PRECONDITION, "cauchy_step: no breakpoint left (the reference indexes fixvars[-1] here, src/basic_tralcnlss.jl:631)");
    if (info_host != 0)
        return fail(BH_ERR_PRECONDITION, "cauchy_step: A_free*A_free' lost positive definiteness (PosDefException in the reference)");
    return BH_OK;
}

int32_t bh_cauchy_step(bh_hess* H, bh_proj* P, const double* x, const double* g, const double* xlow, const double* xupp, double delta,
                       double* s_out, uint64_t* fix_chunks_out, int32_t* n_breakpoints, int32_t* n_hmul_out) {
    return cauchy_impl(H, P, x, g, xlow, xupp, delta, s_out, fix_chunks_out, n_breakpoints, n_hmul_out, false);
}
int32_t bh_cauchy_step_dev(bh_hess* H, bh_proj* P, const double* x_dev, const double* g_dev, const double* xlow_dev, const double* xupp_dev,
                           double delta, double* s_out_dev, uint64_t* fix_chunks_out, int32_t* n_breakpoints, int32_t* n_hmul_out) {
    return cauchy_impl(H, P, x_dev, g_dev, xlow_dev, xupp_dev, delta, s_out_dev, fix_chunks_out, n_breakpoints, n_hmul_out, true);
}

int32_t bh_factor_to_boundary(const double* p, const double* w, const double* w_l, const double* w_u, int64_t n, double atol,
                              double* gamma_out) {
    BH_REQUIRE_INIT();
    if (!p || !w || !w_l || !w_u || !gamma_out || n < 0) return fail(BH_ERR_INVALID_ARG, "bad argument");
    double* buf = nullptr;
    BH_TRY(dev_alloc(&buf, 4 * std::max<int64_t>(n, 1) + 1));
    const double* src[4] = {p, w, w_l, w_u};
    for (int i = 0; i < 4; ++i)
        if (n > 0 && hipMemcpyAsync(buf + i * n, src[i], (size_t)n * sizeof(double), hipMemcpyHostToDevice, g_ctx.stream) != hipSuccess) {
            dev_free(buf);
            return fail(BH_ERR_HIP, "upload");
        }
    hipLaunchKernelGGL(f2b_kernel, dim3(1), dim3(CG_T), 0, g_ctx.stream, buf, buf + n, buf + 2 * n, buf + 3 * n, (int)n, atol, buf + 4 * n);
    hipError_t e = hipMemcpyAsync(gamma_out, buf + 4 * n, sizeof(double), hipMemcpyDeviceToHost, g_ctx.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_ctx.stream);
    dev_free(buf);
    if (e != hipSuccess) return fail(BH_ERR_HIP, std::string("f2b: ") + hipGetErrorString(e));
    return BH_OK;
}

// ---- plumbing -------------------------------------------------------------------------------
int32_t bh_dev_alloc(void** out, int64_t bytes) {
    BH_REQUIRE_INIT();
    if (!out || bytes < 0) return fail(BH_ERR_INVALID_ARG, "bad argument");
    BH_HIP(hipMalloc(out, (size_t)std::max<int64_t>(bytes, 8)));
    return BH_OK;
}
int32_t bh_dev_free(void* p) { dev_free(p); return BH_OK; }
int32_t bh_dev_upload(void* dst_dev, const void* src_host, int64_t bytes) {
    BH_REQUIRE_INIT();
    if (bytes < 0 || (bytes > 0 && (!dst_dev || !src_host))) return fail(BH_ERR_INVALID_ARG, "bad argument");
    if (bytes == 0) return BH_OK;
    count_h2d((size_t)bytes);
    note_dma();
    BH_HIP(hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, g_ctx.stream));
    BH_TRY(sync_flush());
    return BH_OK;
}
int32_t bh_dev_download(void* dst_host, const void* src_dev, int64_t bytes) {
    BH_REQUIRE_INIT();
    if (bytes < 0 || (bytes > 0 && (!dst_host || !src_dev))) return fail(BH_ERR_INVALID_ARG, "bad argument");
    if (bytes == 0) return BH_OK;
    count_d2h((size_t)bytes);
    note_dma();
    BH_HIP(hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, g_ctx.stream));
    BH_TRY(sync_flush());
    return BH_OK;
}

int32_t bh_stats(bh_hess* H, bh_stats_t* out) {
    if (!H || !out) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    *out = H->stats;
    out->h2d_bytes = g_ctx.h2d_bytes; out->d2h_bytes = g_ctx.d2h_bytes;       // library-wide, since bh_init
    out->h2d_calls = g_ctx.h2d_calls; out->d2h_calls = g_ctx.d2h_calls;
    return BH_OK;
}
int32_t bh_stats_reset(bh_hess* H) {
    if (!H) return fail(BH_ERR_INVALID_ARG, "NULL argument");
    const double b = H->stats.bytes_per_hmul;
    H->stats = bh_stats_t{};
    H->stats.bytes_per_hmul = b;
    H->hmul_seq = 0;          // the first H*p after a reset is always sampled (BH_FLAG_PROFILE)
    return BH_OK;
}

int32_t bh_time_kernel(bh_hess* H, int32_t kind, int32_t reps, double* avg_ms) {
    BH_REQUIRE_INIT();
    if (!H || !avg_ms || reps < 1 || kind < 0 || kind > 8) return fail(BH_ERR_INVALID_ARG, "bad argument");
    BH_TRY(hess_ready(H));
    if (kind == 7 && !comm_active()) return fail(BH_ERR_PRECONDITION, "bh_time_kernel(7): no communicator (bh_comm_init; BH_FORCE_COMM=1 for one rank)");
    struct EventPair {          // destroyed on every way out of this function
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } evp;
    BH_HIP(hipEventCreate(&evp.a));
    BH_HIP(hipEventCreate(&evp.b));
    const hipEvent_t e0 = evp.a, e1 = evp.b;
    if (kind >= 7) {
        // 7: the all-reduce of one n-vector on the active communicator path; 8: everything an H*p does after its streaming
        // kernel (slab reduction + exchange).  Back-to-back launches between two events: all ranks must call this together.
        const int cfg = multi_panel(H) ? kPanelCfg : pick_config(H->nchunks);
        const int grid = grid_for(cfg, H->d + H->q_eff);
        int32_t rc = BH_OK;
        for (int i = 0; i < 3 && rc == BH_OK; ++i) rc = (kind == 7) ? allreduce_inplace(H->zpad, H->n, nullptr) : reduce_slabs(H, grid, H->zpad, nullptr);
        if (rc == BH_OK && hipEventRecord(e0, g_ctx.stream) != hipSuccess) rc = fail(BH_ERR_HIP, "hipEventRecord");
        for (int i = 0; i < reps && rc == BH_OK; ++i) rc = (kind == 7) ? allreduce_inplace(H->zpad, H->n, nullptr) : reduce_slabs(H, grid, H->zpad, nullptr);
        if (rc == BH_OK && hipEventRecord(e1, g_ctx.stream) != hipSuccess) rc = fail(BH_ERR_HIP, "hipEventRecord");
        if (rc == BH_OK && hipEventSynchronize(e1) != hipSuccess) rc = fail(BH_ERR_HIP, "hipEventSynchronize");
        float ms = 0.f;
        if (rc == BH_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = fail(BH_ERR_HIP, "hipEventElapsedTime");
        if (rc == BH_OK) rc = check_peer_error();
        if (rc != BH_OK) return rc;
        *avg_ms = ms / reps;
        return BH_OK;
    }
    if (kind >= 3) {
        // read-only stream probe over the (d + q) x ld image: 1, 2, 4 or 8 workgroups per CU
        const int grid = g_ctx.n_cu * (1 << (kind - 3));
        const int64_t nchunks_total = (H->d + H->q) * (H->ld / 2);
        if (grid > H->g_cap * (int)H->ld) return fail(BH_ERR_UNSUPPORTED, "probe output does not fit the slab buffer");
        double total = 0.0;
        for (int i = -1; i < reps; ++i) {           // i = -1: warm-up
            BH_HIP(hipEventRecord(e0, g_ctx.stream));
            hipLaunchKernelGGL(read_probe_kernel, dim3(grid), dim3(256), 0, g_ctx.stream, (const double*)H->Jd, nchunks_total, H->partials);
            BH_HIP(hipEventRecord(e1, g_ctx.stream));
            BH_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            BH_HIP(hipEventElapsedTime(&ms, e0, e1));
            if (i >= 0) total += ms;
        }
        *avg_ms = total / reps;
        return BH_OK;
    }
    if (multi_panel(H)) return fail(BH_ERR_UNSUPPORTED, "bh_time_kernel: single-panel handles only (n <= 16384)");
    const int cfg = pick_config(H->nchunks);
    RowStreamArgs a{};
    a.J = H->Jd; a.ld = H->ld; a.d_rows = H->d; a.nchunks = H->nchunks; a.mu = H->mu; a.state = nullptr;
    a.v = H->vpad; a.u = H->upad; a.partials = H->partials;
    a.nrows = (kind == 0) ? H->d + H->q_eff : H->d;
    a.t_out = (kind == 1) ? H->upad : nullptr;
    const int mode = kind == 0 ? MODE_FUSED : (kind == 1 ? MODE_JV : MODE_JTV);
    const int grid = grid_for(cfg, a.nrows);
    launch_row_stream(cfg, mode, a, grid, g_ctx.stream);   // warm-up
    double total = 0.0;
    for (int i = 0; i < reps; ++i) {
        BH_HIP(hipEventRecord(e0, g_ctx.stream));
        launch_row_stream(cfg, mode, a, grid, g_ctx.stream);
        BH_HIP(hipEventRecord(e1, g_ctx.stream));
        BH_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        BH_HIP(hipEventElapsedTime(&ms, e0, e1));
        total += ms;
    }
    *avg_ms = total / reps;
    return BH_OK;
}

// Self-test of the wave reduction network; returns BH_OK when sum and min agree with a host computation.
int32_t bh_selftest(void) {
    BH_REQUIRE_INIT();
    double host_in[256], host_out[512];
    for (int i = 0; i < 256; ++i) host_in[i] = (double)((i * 37) % 101) - 50.0 + 1.0 / (double)(i + 3);
    note_dma();
    BH_HIP(hipMemcpyAsync(g_ctx.scratch_dev, host_in, sizeof(host_in), hipMemcpyHostToDevice, g_ctx.stream));
    hipLaunchKernelGGL(selftest_wave_kernel, dim3(1), dim3(256), 0, g_ctx.stream, g_ctx.scratch_dev, g_ctx.scratch_dev + 256);
    BH_HIP(hipMemcpyAsync(host_out, g_ctx.scratch_dev + 256, sizeof(host_out), hipMemcpyDeviceToHost, g_ctx.stream));
    BH_TRY(sync_flush());
    for (int w = 0; w < 4; ++w) {
        double mn = host_in[64 * w];
        long double s = 0;
        for (int l = 0; l < 64; ++l) { s += host_in[64 * w + l]; mn = std::min(mn, host_in[64 * w + l]); }
        for (int l = 0; l < 64; ++l) {
            if (std::fabs(host_out[64 * w + l] - (double)s) > 1e-11) return fail(BH_ERR_HIP, "wave_sum self-test mismatch");
            if (host_out[64 * w + l] != host_out[64 * w]) return fail(BH_ERR_HIP, "wave_sum lanes disagree");
            if (host_out[256 + 64 * w + l] != mn) return fail(BH_ERR_HIP, "wave_min self-test mismatch");
        }
    }
    // tri_inv_small_kernel (the explicit inverse behind the three-kernel equality iteration): L Linv = I, Linv' stored
    // transposed, zeros above the diagonal and beyond m — every block boundary of the recursion (16 / 32 / 64) and odd sizes
    double *Ld = nullptr, *Wd = nullptr;
    BH_TRY(dev_alloc(&Ld, 64 * 64 + 64));
    int32_t rc = dev_alloc(&Wd, 2 * 4096);
    std::vector<double> L(64 * 64 + 64), W(2 * 4096);
    for (int m : {1, 2, 7, 15, 16, 17, 31, 32, 33, 48, 63, 64}) {
        if (rc != BH_OK) break;
        std::fill(L.begin(), L.end(), std::nan(""));                    // the upper triangle must never be read
        for (int k = 0; k < m; ++k)
            for (int i = k; i < m; ++i)
                L[(size_t)i + (size_t)k * m] = (i == k) ? 1.5 + 0.01 * i : 0.3 * std::sin(1.0 + 0.7 * i + 1.3 * k) / (1.0 + 0.1 * (i - k));
        for (int i = 0; i < m; ++i) L[(size_t)m * m + i] = 1.0 / L[(size_t)i + (size_t)i * m];
        if (hipMemcpy(Ld, L.data(), ((size_t)m * m + m) * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(BH_ERR_HIP, "self-test upload"); break; }
        hipLaunchKernelGGL(tri_inv_small_kernel, dim3(1), dim3(256), 0, g_ctx.stream, (const double*)Ld, m, Wd);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(g_ctx.stream) != hipSuccess ||
            hipMemcpy(W.data(), Wd, W.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(BH_ERR_HIP, "tri_inv self-test launch"); break; }
        for (int i = 0; i < 64 && rc == BH_OK; ++i)
            for (int k = 0; k < 64; ++k) {
                const double x = W[(size_t)k * 64 + i];                 // Linv[i][k]
                if (W[4096 + (size_t)i * 64 + k] != x) { rc = fail(BH_ERR_HIP, "tri_inv self-test: the transposed copy differs"); break; }
                if ((i >= m || k >= m || k > i) && x != 0.0) { rc = fail(BH_ERR_HIP, "tri_inv self-test: non-zero outside the triangle"); break; }
                if (i < m && k <= i) {
                    long double acc = 0;                                 // (L Linv)[i][k]
                    for (int l = k; l <= i; ++l) acc += (long double)L[(size_t)i + (size_t)l * m] * (long double)W[(size_t)k * 64 + l];
                    if (std::fabs((double)acc - (i == k ? 1.0 : 0.0)) > 1e-13) { rc = fail(BH_ERR_HIP, "tri_inv self-test: L Linv != I at m = " + std::to_string(m)); break; }
                }
            }
    }
    dev_free(Ld); dev_free(Wd);
    return rc;
}

}  // extern "C"

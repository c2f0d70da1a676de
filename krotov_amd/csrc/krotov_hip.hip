// libkrotov_hip.so -- C ABI of the MI355X Krotov engine (include/krotov_hip.h).
//
// Host side: engine object, operator staging (adjoint copies), kernel-family
// selection and launches.  Device side: kh_tile64.h (N <= 64, operators in
// registers) and kh_generic.h (any N).  gfx950 only.
#include <hip/hip_runtime.h>

#include <cxxabi.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <climits>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/krotov_hip.h"
#include "kh_common.h"
#include "kh_generic.h"
#include "kh_tile64.h"
#include "kh_tile64s.h"
#include "kh_tile64x.h"
#include "kh_tile64q2.h"
#ifdef KH_WITH_Q4  // experiment build only (scripts/experiments/kh_tile64q4.h: 1024-thread plain sweep, measured 71 % slower)
#include "../../scripts/experiments/kh_tile64q4.h"
#endif
#include "kh_coop.h"
#ifdef KH_WITH_C4W  // experiment build only (scripts/experiments/kh_coop4w.h: rows over waves, measured slower)
#include "../../scripts/experiments/kh_coop4w.h"
#endif
#include "kh_mini.h"
#include "kh_ell.h"
#include "kh_tilen.h"
#include "kh_ens.h"
// the parallel build (krotov_amd/build.py, -DKH_TU=KH_TU_MAIN): the sweep kernels are instantiated in the family units
// (kh_tu.hip), here they are `extern template`; compiled by itself this file is the whole library in one unit
#include "kh_instances.inc"

static thread_local std::string g_last_error;

static int kh_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define KH_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t _e = (call);                                                               \
        if (_e != hipSuccess)                                                                 \
            return kh_fail(KH_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), \
                           __FILE__, __LINE__);                                               \
    } while (0)

enum KernelKind { KIND_GENERIC = 0, KIND_TILE_RPT2 = 1, KIND_TILE_RPT1 = 2, KIND_TILE_Q2 = 3, KIND_COOP = 4, KIND_ELL = 5, KIND_TILEN = 6, KIND_TILEX = 7 /* plain sweeps only: kind_store */ };

struct kh_engine {
    int K, N, L, nt, is_super;
    double tol, theta_max;
    int device, num_cus;
    KernelKind kind;
    KernelKind kind_store;  // family of the plain sweeps (no cross-objective coupling: any K)
    int grid_update;  // workgroups of the single-launch update sweep
    // cooperative shared-operator kernels (kh_coop.h): row blocks, column groups, k-steps per wave
    int coop_G = 0, coop_Y = 0, coop_ks = 0, coop_cols = KH_COOP_COLS;
    kh_u64 *d_coop_vbuf = nullptr;
    unsigned int *d_coop_xcc = nullptr;  // [Y * G] placement check of the cooperative kernels (zeroed per launch)
    bool coop_xcd = false;               // one column group per XCD (kh_coop.h, kh_coop_place); KH_COOP_XCD=0: off
    size_t coop_vbuf_bytes = 0;
    // update sums on the adjoint side (kh_coop_adjoint_side): H_1^+ chi for the whole store, formed in front of the
    // update sweep; KH_COOP_NO_ADJ=1: the sums by one more round per interval, as for second order / two controls
    // experiment (-DKH_WITH_C4W builds, KH_COOP4W=1; scripts/experiments/kh_coop4w.h): four waves per workgroup, a wave
    // owns four rows for the whole k range; plain sweeps only; measured slower than the k-split kernels
    bool coop4w = false;
    int c4_NG = 0;  // groups of 16 columns
    const cplx **d_c4_fops_fw = nullptr, **d_c4_fops_bw = nullptr;  // [2] H0, H1 in kh_coop4w.h's fragment order
    const cplx **d_c4_sq_fw = nullptr, **d_c4_sq_bw = nullptr;      // [3] P0, P1, P2
    bool coop_adj = false;
    unsigned char *d_coop_adj_nz = nullptr;  // [G][G] non-zero 16 x 16 blocks of H_1^+
    cplx *d_coop_adj = nullptr;              // [K][nt][N], allocated by the first update sweep
    const cplx *coop_adj_op = nullptr;       // H_1^+, row-major (the staged adjoint of the shared control operator)
    // device-side problem data
    const cplx **d_ops_fw = nullptr;  // [K*(1+L)]
    const cplx **d_ops_bw = nullptr;  // [K*(1+L)] adjoints
    double *d_norms = nullptr;        // [K*(1+L)]
    double *d_dt = nullptr;           // [nt-1]
    double *d_deg_theta = nullptr;    // [KH_MAX_DEGREE+1] degree thresholds for tol
    KhCsr *d_csr_fw = nullptr;        // [K*(1+L)] sparse operators (kh_engine_create_csr), else NULL
    KhCsr *d_csr_bw = nullptr;        // [K*(1+L)] their conjugate transposes
    KhEll *d_ell_fw = nullptr, *d_ell_bw = nullptr;  // [K] sparse operators in padded row form (kh_ell.h), or NULL
    int *d_ell_off = nullptr;         // ... their column offsets and values (one pool each)
    cplx *d_ell_vals = nullptr;
    int ell_E = 0;                    // widest row over all objectives and both directions (picks the instantiation)
    bool gen_fits = true;             // the generic kernels' LDS vectors fit (N <= 2540)
    bool ell_stream = false;          // the streamed form of kh_ell.h (rows that do not fit the registers, N <= 4096)
    cplx *d_ell_scratch = nullptr;    // ... its per-workgroup scratch planes [workgroups][ell_scratch_stride]
    long long ell_scratch_stride = 0;
    const cplx **d_coop_fops_fw = nullptr, **d_coop_fops_bw = nullptr;  // [1+L] fragment-ordered operator copies
    const cplx **d_coop_sq_fw = nullptr, **d_coop_sq_bw = nullptr;      // [3] the same for P0, P1, P2 (one control)
    const cplx **d_sq_fw = nullptr;   // [K*3] P0, P1, P2 of A^2 (q2 kernels), forward operators
    const cplx **d_sq_bw = nullptr;   // [K*3] the same for the adjoint operators
    const cplx **d_tn_fw = nullptr, **d_tn_bw = nullptr;  // [K*(1+L)] lane-order operator copies (kh_tilen.h), or NULL
    const cplx **d_tx_fw = nullptr, **d_tx_bw = nullptr;  // [K*(1+L)] lane-order 64 x 64 operator copies (kh_tile64x.h), or NULL
    bool tx_update = false;           // ... and the update sweep may take that family too (K <= co-resident workgroups)
    bool tn_h1reg = false;            // ... one control, present in every objective, N <= 96: it stays in registers too
    std::vector<void *> owned;        // adjoint operator copies
    // workspaces
    cplx *d_phi = nullptr;            // [K][N]
    kh_u64 *d_slots = nullptr;        // [2][G][L][2]
    unsigned int *d_abort = nullptr;
    unsigned long long *d_wait_ticks = nullptr;  // [4] kh_p2p_stats: in-GPU gather, cross-GPU wait (last sharded sweep); self-test ticks, rounds
    double *d_stats = nullptr;        // [4] (+ 64 trace stamps behind them in a KH_TIMING build)
    double *d_wg_partial = nullptr;   // [G][L]
    cplx *d_gen_scratch = nullptr;    // [gen_scratch_wgs][N][N] generic kernels: the interval's generator (ensure_gen_scratch)
    // generic kernels, first order, dense operators: the update sums on the adjoint side (kh_gen_adjoint_side):
    // H_lk^+ chi_k(t_n) for the whole co-state store, [L][K][nt][N], formed in front of every update sweep
    cplx *d_gen_adj = nullptr;
    bool gen_adj_failed = false;      // the allocation did not fit: the sums stay on the forward side
    bool gen_adj_ready = false;       // d_gen_adj holds the store of the sweep in progress (stepwise launches reuse it)
    int gen_scratch_wgs = 0;
    bool gen_scratch_failed = false;
    const double *guess_dev = nullptr;  // remembered by kh_update_begin
    // second-order update (kh_set_second_order); all NULL = first order
    const cplx *so_fw_prev = nullptr;
    cplx *so_fw_store = nullptr;
    const double *so_sigma = nullptr;
    // cross-GPU exchange (kh_p2p_*): objectives sharded over `p2p_world` ranks
    int p2p_world = 1, p2p_rank = 0;
    kh_u64 *p2p_window = nullptr;            // this rank's window (fine-grained device memory)
    size_t p2p_window_bytes = 0;
    std::vector<void *> p2p_opened;          // peer windows opened through IPC
    kh_u64 **d_p2p_peers = nullptr;          // device array [world] of window pointers
    unsigned int p2p_epoch_base = 0;         // advanced by nt per sweep: epochs never repeat
    bool p2p_ready = false;
    size_t slots_bytes = 0;
    double last_intervals = 0, last_wgs = 0;
    std::set<const void *> lds_raised;  // kernels whose dynamic-LDS limit was raised on this engine's device
    // tuning knobs (s_sleep units of 64 cycles), read from the environment once at creation
    int poll_delay = 16;       // KH_POLL_DELAY: head start of the update-sum stores, ~0.4 us: measured best (one control)
    bool poll_delay_set = false;  // ... given in the environment: several controls do not apply their own default then
    int adj_poll_delay = 0;    // KH_ADJ_DELAY: the same where a matrix-vector product already sits between store and poll
    int coop_poll_delay = 0;   // KH_COOP_DELAY: the same for the cooperative kernels' block exchange (a polling pass is four
                               // 16-byte loads per lane there: an early, stale pass costs little -- measured 0 best)
    double adj_sign = 0.0;  // +1 / -1: every control operator equals +/- its adjoint exactly (else 0)
    bool real_spectrum = false;  // every operator Hermitian (bit for bit) and f = -+i
    double imag_defect = -1.0;   // >= 0: bound on the Hermitian part of f A dt when the controls' f H_l are exactly
                                 // anti-Hermitian (|| . ||_F of the drift's part x max dt); < 0: not of that kind
    bool coop_series = false;    // the cooperative kernels run the Chebyshev-form series (kh_common.h)
    bool use_q4 = false;         // KH_Q4=1 in a -DKH_WITH_Q4 build (experiment): plain sweeps with 1024-thread workgroups
    bool stepwise_only = false;  // more objectives than can be co-resident: kh_forward_update runs one launch per interval
    // ... unless the streaming kernel takes them (kh_tile64s.h): ONE launch of stream_G co-resident workgroups, each
    // walking through its objectives in every interval (one GPU; KH_NO_STREAM=1: the launch per interval, A/B)
    bool stream = false;
    int stream_G = 0;
    int last_update_grid = 0;  // workgroups of the last single-launch update sweep where the ensemble / streaming kernels ran it
    int reduced_G = 0;  // kh_set_update_workgroups: the single-launch update sweep on at most this many workgroups (0: off)
    double *d_step_partial = nullptr;  // [L] the interval's sums on that path
    // ensembles (kh_ens.h): every objective's operator list is (H0, s_k H1) with one H0 and one H1 -- the single-launch
    // update sweep then runs on the matrix cores with 2 ens_ncg objectives per workgroup, whatever `kind` says (which
    // still serves the plain sweeps and the per-interval form)
    bool ens = false;
    int ens_ncg = 0, ens_G = 0;
    const cplx *ens_H0 = nullptr, *ens_H1 = nullptr;
    bool ens2 = true;  // first order, 4 objectives per workgroup: the A^2-chain form (kh_ens2_forward_update); KH_ENS2=0: off
    double *d_ens_scale = nullptr;     // [K]
    bool mini = false;           // kind q2, N <= 16, K <= 8: the one-wave-per-objective kernels (kh_mini.h)
    bool quad = false;           // mini with N <= 4, K <= 4: the whole problem in one wave
    double *d_q2_theta = nullptr, *d_q2_c0 = nullptr, *d_q2_rows = nullptr, *d_ratios = nullptr;  // series tables of the register-tile kernels
    long long timeout_ticks = 100000000LL;  // KH_TIMEOUT_MS: bound on any in-kernel wait (100 MHz ticks; 1 s)
    bool timeout_set = false;    // ... given in the environment (the cross-GPU sweeps then keep it instead of their 10 s)
    // switches read from the environment ONCE, at creation (not per launch, not per process: engines with different
    // settings coexist)
    bool coop_launch = true;     // KH_COOP_LAUNCH=0: plain launches instead of cooperative ones (A/B timing)
    bool q2_single = true;       // KH_Q2_SINGLE=0: the instantiations with the cross-GPU stage on one GPU too (A/B)
    bool tile_single = true;     // KH_TILE_SINGLE=0: the same for the one-term-per-phase kernels
    bool coop_single = true;     // KH_COOP_SINGLE=0: the same for the cooperative kernels' adjoint-side form
    // fault injection for the sharded protocol (tests): rank KH_P2P_FAIL_RANK withholds its GPU's sum at interval
    // KH_P2P_FAIL_AT of its KH_P2P_FAIL_SWEEP-th update sweep through the peer windows (1-based; default 1)
    int p2p_fail_at = -1, p2p_fail_rank = 0, p2p_fail_sweep = 1, p2p_sweeps = 0;
};

// Kernels with more than 64 KiB of dynamic LDS need the limit raised once per device: remembered per
// engine (an engine is bound to one device), not per process.
static int ensure_dynamic_lds(kh_engine *e, const void *func, size_t bytes) {
    if (bytes <= 48 * 1024 || e->lds_raised.count(func)) return KH_OK;
    KH_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    e->lds_raised.insert(func);
    return KH_OK;
}

// The update-sweep kernels that exchange partial sums in-kernel spin until EVERY workgroup of the grid has
// published: all of them must be resident at once.  Launching them cooperatively makes the runtime check the grid
// against the occupancy of this very kernel (register / LDS footprint as built) -- instead of assuming one
// workgroup per CU from multiProcessorCount.  It does NOT keep other streams off the device (ROCm 7.2, measured:
// tests/test_hip_parity.py::test_update_sweep_next_to_a_busy_stream): CUs held by somebody else still lead to a
// partial start, which the bounded in-kernel waits turn into KH_ERR_TIMEOUT and the caller into a repeat of the
// sweep with one launch per interval (krotov_amd/optimize.py).
// KH_COOP_LAUNCH=0 keeps plain launches (A/B timing: a cooperative launch costs ~15-20 us of host time).
// ---- which sweep-kernel instantiations exist and which have been launched (kh_debug_launched) ----
// Every launch of a sweep kernel goes through launch_plain<Kernel> / launch_persistent<Kernel>; naming the kernel as a
// template argument instantiates KhKernelTag<Kernel>, whose static member registers the instantiation when the library
// is loaded: the registry is exactly the set of instantiations some dispatch can select.
// (engines may be driven from different host threads -- one engine per thread, INTEGRATION.md 4 --: the flags are atomics,
// the registry itself is only appended to while the library is loaded)
struct KhKernelRecord {
    std::string name;
    std::atomic<bool> launched{false};
    std::atomic<bool> logged{false};  // written to KH_LAUNCH_LOG by this process
    explicit KhKernelRecord(std::string n) : name(std::move(n)) {}
};
static std::deque<KhKernelRecord> &kh_kernel_registry() {
    static std::deque<KhKernelRecord> reg;
    return reg;
}
static int kh_register_kernel(const void *host_stub) {
    // the host-side launch stub's symbol, demangled: "void __device_stub__kh_q2_forward_update<false, true, true>(KhSweepArgs, ...)"
    std::string s = "?";
    Dl_info info;
    if (dladdr(host_stub, &info) != 0 && info.dli_sname != nullptr) {
        int status = 0;
        char *dem = abi::__cxa_demangle(info.dli_sname, nullptr, nullptr, &status);
        s = (status == 0 && dem != nullptr) ? dem : info.dli_sname;
        free(dem);
        if (s.compare(0, 5, "void ") == 0) s = s.substr(5);
        const size_t stub = s.find("__device_stub__");
        if (stub != std::string::npos) s.erase(stub, 15);
        // cut the parameter list: the '(' that closes the name (template arguments hold no parentheses here)
        const size_t paren = s.find('(');
        if (paren != std::string::npos) s = s.substr(0, paren);
    }
    kh_kernel_registry().emplace_back(s);
    return (int)kh_kernel_registry().size() - 1;
}
template <auto Kernel>
struct KhKernelTag {
    static inline const int index = kh_register_kernel((const void *)Kernel);
};
static void kh_note_launch(int index) {
    KhKernelRecord &rec = kh_kernel_registry()[index];
    rec.launched.store(true, std::memory_order_relaxed);
    if (rec.logged.load(std::memory_order_relaxed)) return;
    // (tests: one line per instantiation and process, appended; the variable is read at every launch so that a test
    // session can switch the log on for its oracle-comparing tests only -- tests/conftest.py)
    if (const char *path = getenv("KH_LAUNCH_LOG")) {
        if (path[0] == 0) return;
        if (FILE *f = fopen(path, "a")) {
            fprintf(f, "%s\n", rec.name.c_str());
            fclose(f);
            rec.logged.store(true, std::memory_order_relaxed);
        }
    }
}
template <class... P>
static std::tuple<P...> kh_param_tuple(void (*)(P...));

template <auto Kernel, class... Args>
static void launch_plain(dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    kh_note_launch(KhKernelTag<Kernel>::index);
    hipLaunchKernelGGL(Kernel, grid, block, lds, st, args...);
}

template <auto Kernel, class... Args>
static int launch_persistent(const kh_engine *e, dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    kh_note_launch(KhKernelTag<Kernel>::index);
    if (grid.x * grid.y * grid.z == 1) {
        hipLaunchKernelGGL(Kernel, grid, block, lds, st, args...);
        return KH_OK;
    }
    if (!e->coop_launch) {
        // a plain launch has the same residency but nobody checks the grid against it (the cooperative launch below
        // does): ask the occupancy of THIS instantiation as built -- registers, scratch, LDS -- once per shape, so that
        // a grid that cannot be co-resident is refused here (KH_ERR_UNSUPPORTED: the caller takes a smaller grid or one
        // launch per interval) instead of ending in a timeout
        static std::map<std::tuple<const void *, unsigned, size_t, int>, int> per_cu_of;
        static std::mutex per_cu_lock;  // (two engines launching from two host threads share the cache)
        const auto key = std::make_tuple((const void *)Kernel, block.x, lds, e->device);
        int per_cu = -1;
        {
            std::lock_guard<std::mutex> hold(per_cu_lock);
            auto it = per_cu_of.find(key);
            if (it != per_cu_of.end()) per_cu = it->second;
        }
        if (per_cu < 0) {
            per_cu = 0;
            KH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)Kernel, (int)block.x, lds));
            std::lock_guard<std::mutex> hold(per_cu_lock);
            per_cu_of.emplace(key, per_cu);
        }
        if ((long long)per_cu * e->num_cus < (long long)grid.x * grid.y * grid.z)
            return kh_fail(KH_ERR_UNSUPPORTED, "the update sweep's %u workgroups cannot all be resident on this device (%d per CU)",
                           grid.x * grid.y * grid.z, per_cu);
        hipLaunchKernelGGL(Kernel, grid, block, lds, st, args...);
        return KH_OK;
    }
    decltype(kh_param_tuple(Kernel)) packed(args...);
    constexpr size_t NP = std::tuple_size<decltype(packed)>::value;
    void *ptrs[NP];
    int i = 0;
    std::apply([&](auto &...a) { ((ptrs[i++] = (void *)&a), ...); }, packed);
    const hipError_t err = hipLaunchCooperativeKernel((const void *)Kernel, grid, block, ptrs, (unsigned int)lds, st);
    if (err == hipErrorCooperativeLaunchTooLarge) {
        (void)hipGetLastError();
        return kh_fail(KH_ERR_UNSUPPORTED, "the update sweep's %u workgroups cannot all be resident on this device",
                       grid.x * grid.y * grid.z);
    }
    if (err != hipSuccess) return kh_fail(KH_ERR_HIP, "hipLaunchCooperativeKernel failed: %s", hipGetErrorString(err));
    return KH_OK;
}

// grid <= (resident workgroups per CU of THIS kernel) x CUs ?  (checked once per kernel at engine creation)
static int check_residency(const kh_engine *e, const void *func, int threads, size_t lds, int grid, const char *what);

extern "C" const char *kh_last_error(void) { return g_last_error.c_str(); }

static int check_residency(const kh_engine *e, const void *func, int threads, size_t lds, int grid, const char *what) {
    int per_cu = 0;
    KH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, func, threads, lds));
    if ((long long)per_cu * e->num_cus < grid)
        return kh_fail(KH_ERR_UNSUPPORTED, "%s: %d workgroups needed at once, %d x %d CUs resident", what, grid, per_cu,
                       e->num_cus);
    return KH_OK;
}

extern "C" const char *kh_version(void) { return "krotov_hip 0.6 (gfx950; tile64q2, tile64, tile64/stream, tile64x, ens64/mfma, mini16, mini4, coop16/mfma, ell/csr, ellstream/csr, tile128, generic, generic/csr kernels)"; }

extern "C" const char *kh_engine_kernel(const kh_engine *e) {
    if (e == nullptr) return "";
    if (e->ens) return "ens64/mfma";
    switch (e->kind) {
        case KIND_TILE_RPT2: return "tile64/256";
        case KIND_TILE_RPT1: return e->stepwise_only ? (e->stream ? "tile64/stream" : "tile64/512 per interval") : "tile64/512";
        case KIND_TILE_Q2: return e->mini ? (e->quad ? "mini4/wave" : "mini16/wave") : "tile64q2/512";
        case KIND_COOP: return "coop16/mfma";
        case KIND_ELL: return e->ell_stream ? "ellstream/csr" : "ell/csr";
        case KIND_TILEN: return "tile128/512";
        default: return e->d_csr_fw != nullptr ? "generic/csr" : (e->d_tx_fw != nullptr ? "tile64x/512" : "generic");
    }
}

static KhSweepArgs sweep_args(const kh_engine *e, bool backward) {
    KhSweepArgs p;
    p.K = e->K;
    p.N = e->N;
    p.L = e->L;
    p.nt = e->nt;
    p.ops = backward ? e->d_ops_bw : e->d_ops_fw;
    p.csr = backward ? e->d_csr_bw : e->d_csr_fw;
    p.op_norms = e->d_norms;
    p.dt = e->d_dt;
    // equation-of-motion factor (propagators.py:94-99): -i, conj for backwards; 1 for Liouvillians
    if (e->is_super) {
        p.fre = 1.0;
        p.fim = 0.0;
    } else {
        p.fre = 0.0;
        p.fim = backward ? 1.0 : -1.0;
    }
    p.tol = e->tol;
    p.theta_max = e->theta_max;
    p.inv_theta_max = 1.0 / e->theta_max;
    p.deg_theta = e->d_deg_theta;
    p.q2_theta = e->d_q2_theta;
    p.q2_c0 = e->d_q2_c0;
    p.q2_rows = e->d_q2_rows;
    p.ratios = e->d_ratios;
    p.stats = e->d_stats;
    p.gen_scratch = e->d_gen_scratch;
    p.gen_scratch_wgs = e->gen_scratch_wgs;
    return p;
}

// The generic kernels' scratch generators (kh_generic.h: dense operators with N > 96): one N x N matrix per workgroup,
// allocated at the first launch that can use it; at most 4 GiB: a launch that asks for more workgroups than that holds
// gets as many scratch matrices as fit (the kernel forms the generator for blockIdx.x < gen_scratch_wgs and streams the
// operators per term in the others), and only a failed allocation leaves the streamed form for good.
static void ensure_gen_scratch(kh_engine *e, int wgs) {
    if (e->d_csr_fw != nullptr || kh_gen_lds_A(e->N, true) || e->gen_scratch_failed) return;
    const size_t per_wg = sizeof(cplx) * (size_t)e->N * e->N;
    const size_t fit = ((size_t)4 << 30) / per_wg;
    if ((size_t)wgs > fit) wgs = (int)fit;
    if (wgs < 1 || e->gen_scratch_wgs >= wgs) return;
    if (e->d_gen_scratch != nullptr) (void)hipFree(e->d_gen_scratch);
    e->d_gen_scratch = nullptr;
    e->gen_scratch_wgs = 0;
    if (hipMalloc(&e->d_gen_scratch, per_wg * (size_t)wgs) != hipSuccess) {
        (void)hipGetLastError();
        e->d_gen_scratch = nullptr;
        e->gen_scratch_failed = true;
        return;
    }
    e->gen_scratch_wgs = wgs;
}

// The generic kernels' adjoint-side store [L][K][nt][N]: allocated at the first update sweep that can use it, at most
// a quarter of the device memory that is free then (KH_GEN_ADJ=0: never); a store that does not fit leaves the sums on the
// forward side for good.
static bool ensure_gen_adj(kh_engine *e) {
    if (e->d_gen_adj != nullptr) return true;
    if (e->gen_adj_failed) return false;
    if (const char *d = getenv("KH_GEN_ADJ"))
        if (atoi(d) == 0) {
            e->gen_adj_failed = true;
            return false;
        }
    const size_t bytes = sizeof(cplx) * (size_t)e->L * e->K * e->nt * e->N;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes > free_b / 4 || hipMalloc(&e->d_gen_adj, bytes) != hipSuccess) {
        (void)hipGetLastError();
        e->d_gen_adj = nullptr;
        e->gen_adj_failed = true;
        return false;
    }
    return true;
}

extern "C" void kh_engine_destroy(kh_engine *e) {
    if (e == nullptr) return;
    for (void *ptr : e->owned) (void)hipFree(ptr);
    (void)hipFree((void *)e->d_ops_fw);
    (void)hipFree((void *)e->d_ops_bw);
    (void)hipFree(e->d_norms);
    (void)hipFree(e->d_dt);
    (void)hipFree(e->d_deg_theta);
    (void)hipFree(e->d_q2_theta);
    (void)hipFree(e->d_q2_c0);
    (void)hipFree(e->d_q2_rows);
    (void)hipFree(e->d_ratios);
    (void)hipFree(e->d_csr_fw);
    (void)hipFree(e->d_csr_bw);
    (void)hipFree(e->d_ell_fw);
    (void)hipFree(e->d_ell_bw);
    (void)hipFree(e->d_ell_off);
    (void)hipFree(e->d_ell_vals);
    (void)hipFree((void *)e->d_coop_fops_fw);
    (void)hipFree((void *)e->d_coop_fops_bw);
    (void)hipFree((void *)e->d_coop_sq_fw);
    (void)hipFree((void *)e->d_coop_sq_bw);
    (void)hipFree((void *)e->d_c4_fops_fw);
    (void)hipFree((void *)e->d_c4_fops_bw);
    (void)hipFree((void *)e->d_c4_sq_fw);
    (void)hipFree((void *)e->d_c4_sq_bw);
    (void)hipFree((void *)e->d_sq_fw);
    (void)hipFree((void *)e->d_sq_bw);
    (void)hipFree((void *)e->d_tn_fw);
    (void)hipFree((void *)e->d_tn_bw);
    (void)hipFree((void *)e->d_tx_fw);
    (void)hipFree((void *)e->d_tx_bw);
    (void)hipFree(e->d_phi);
    (void)hipFree(e->d_slots);
    (void)hipFree(e->d_abort);
    (void)hipFree(e->d_wait_ticks);
    (void)hipFree(e->d_gen_scratch);
    (void)hipFree(e->d_gen_adj);
    (void)hipFree(e->d_ell_scratch);
    (void)hipFree(e->d_stats);
    (void)hipFree(e->d_wg_partial);
    (void)hipFree(e->d_step_partial);
    (void)hipFree(e->d_ens_scale);
    (void)hipFree(e->d_coop_vbuf);
    (void)hipFree(e->d_coop_xcc);
    (void)hipFree(e->d_coop_adj_nz);
    (void)hipFree(e->d_coop_adj);
    for (void *ptr : e->p2p_opened) (void)hipIpcCloseMemHandle(ptr);
    (void)hipFree(e->p2p_window);
    (void)hipFree((void *)e->d_p2p_peers);
    delete e;
}

// ---------------------------------------------------------------------------
// sparse operators: host-side analysis and the padded row form of kh_ell.h
// ---------------------------------------------------------------------------
struct HostCsr {  // canonical: column indices sorted within a row, duplicates summed, explicit zeros dropped
    std::vector<int> indptr, indices;
    std::vector<cplx> data;
};

static bool canonical_csr(const int *indptr, const int *indices, const cplx *data, long long nnz, int N, HostCsr &out);

static hipError_t fetch_csr(const kh_csr &c, int N, HostCsr &out) {
    std::vector<int> indptr(N + 1), indices((size_t)c.nnz);
    std::vector<cplx> data((size_t)c.nnz);
    hipError_t err = hipMemcpy(indptr.data(), c.indptr, sizeof(int) * (N + 1), hipMemcpyDeviceToHost);
    if (err == hipSuccess && c.nnz > 0) err = hipMemcpy(indices.data(), c.indices, sizeof(int) * (size_t)c.nnz, hipMemcpyDeviceToHost);
    if (err == hipSuccess && c.nnz > 0) err = hipMemcpy(data.data(), c.data, sizeof(cplx) * (size_t)c.nnz, hipMemcpyDeviceToHost);
    if (err != hipSuccess) return err;
    return canonical_csr(indptr.data(), indices.data(), data.data(), c.nnz, N, out) ? hipSuccess : hipErrorInvalidValue;
}

// host arrays -> canonical form; false on inconsistent arrays
static bool canonical_csr(const int *indptr, const int *indices, const cplx *data, long long nnz, int N, HostCsr &out) {
    out.indptr.assign(N + 1, 0);
    out.indices.clear();
    out.data.clear();
    std::vector<std::pair<int, cplx>> row;
    for (int r = 0; r < N; ++r) {
        row.clear();
        const int lo = indptr[r], hi = indptr[r + 1];
        if (lo < 0 || hi < lo || hi > nnz) return false;
        for (int j = lo; j < hi; ++j) {
            if (indices[j] < 0 || indices[j] >= N) return false;
            row.emplace_back(indices[j], data[j]);
        }
        std::stable_sort(row.begin(), row.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
        for (size_t j = 0; j < row.size();) {
            cplx v = row[j].second;
            size_t q = j + 1;
            for (; q < row.size() && row[q].first == row[j].first; ++q) {
                v.x += row[q].second.x;
                v.y += row[q].second.y;
            }
            if (v.x != 0.0 || v.y != 0.0) {
                out.indices.push_back(row[j].first);
                out.data.push_back(v);
            }
            j = q;
        }
        out.indptr[r + 1] = (int)out.indices.size();
    }
    return true;
}

// does b equal sign * a, entry for entry?
static bool csr_equal(const HostCsr &a, const HostCsr &b, double sign) {
    if (a.indptr != b.indptr || a.indices != b.indices) return false;
    for (size_t j = 0; j < a.data.size(); ++j)
        if (b.data[j].x != sign * a.data[j].x || b.data[j].y != sign * a.data[j].y) return false;
    return true;
}

// || (a + sign adj) / 2 ||_F^2 with adj the conjugate transpose of a, as supplied by the caller
static double csr_part_fro2(const HostCsr &a, const HostCsr &adj, double sign) {
    double acc = 0.0;
    const int N = (int)a.indptr.size() - 1;
    for (int r = 0; r < N; ++r) {
        int i = a.indptr[r], j = adj.indptr[r];
        const int ie = a.indptr[r + 1], je = adj.indptr[r + 1];
        while (i < ie || j < je) {
            double re = 0.0, im = 0.0;
            const int ci = i < ie ? a.indices[i] : INT32_MAX, cj = j < je ? adj.indices[j] : INT32_MAX;
            if (ci <= cj) {
                re += a.data[i].x;
                im += a.data[i].y;
            }
            if (cj <= ci) {
                re += sign * adj.data[j].x;
                im += sign * adj.data[j].y;
            }
            if (ci <= cj) ++i;
            if (cj <= ci) ++j;
            acc += 0.25 * (re * re + im * im);
        }
    }
    return acc;
}

// One operator list (drift + L controls, canonical host copies; NULL: absent) in the padded row form of kh_ell.h:
// the union of the patterns, entries some control touches first.  Returns false when a row is wider than the kernels'
// register budget (kh_ell_emax(N): 32 entries with one row per lane, 16 with two, 8 with three or four).
static bool build_ell_host(const std::vector<const HostCsr *> &ops, int N, std::vector<int> &off, std::vector<cplx> &vals,
                           int &E, int &Ec, bool stream = false) {
    const int Lp1 = (int)ops.size();
    std::vector<std::vector<std::pair<int, int>>> rows(N);  // (column, touched by a control)
    E = Ec = 0;
    std::map<int, int> cols;
    for (int r = 0; r < N; ++r) {
        cols.clear();
        for (int o = 0; o < Lp1; ++o) {
            if (ops[o] == nullptr) continue;
            for (int j = ops[o]->indptr[r]; j < ops[o]->indptr[r + 1]; ++j) {
                int &flag = cols[ops[o]->indices[j]];
                if (o > 0) flag = 1;
            }
        }
        int nc = 0;
        for (const auto &kv : cols)
            if (kv.second) rows[r].emplace_back(kv.first, 1), ++nc;
        for (const auto &kv : cols)
            if (!kv.second) rows[r].emplace_back(kv.first, 0);
        E = std::max(E, (int)rows[r].size());
        Ec = std::max(Ec, nc);
    }
    // (stream: the pools of the streamed kernels -- nothing lives in registers, so rows up to 32 entries for any N they take)
    const int emax = stream ? KH_ELL_EMAX : kh_ell_emax(N), S = stream ? (N + 63) / 64 * 64 : kh_ell_rows(N);
    if (E > emax) return false;
    // every row: its control-touched entries in slots [0, Ec), the others behind them from slot Ec on (so that a rebuild
    // of slots [0, Ec) never touches a drift-only entry); padding: value 0, the lane's own row
    int width = 0;
    for (int r = 0; r < N; ++r) {
        int nc = 0;
        for (const auto &cv : rows[r]) nc += cv.second;
        width = std::max(width, Ec + ((int)rows[r].size() - nc));
    }
    if (width > emax) return false;
    E = (std::max(width, 1) + 3) / 4 * 4;    // the kernels work on groups of four entries
    const int Ec_true = Ec;
    Ec = (Ec + 3) / 4 * 4;                   // (slots [Ec_true, Ec): drift-only entries or padding -- rebuilt to themselves)
    off.assign((size_t)E * S, 0);
    vals.assign((size_t)Lp1 * E * S, make_double2(0.0, 0.0));
    for (int t = 0; t < S; ++t)
        for (int e = 0; e < E; ++e) off[(size_t)e * S + t] = (t < N ? t : 0) * (int)sizeof(cplx);
    auto value_at = [](const HostCsr *m, int r, int c, cplx &v) {
        if (m == nullptr) return false;
        const auto lo = m->indices.begin() + m->indptr[r], hi = m->indices.begin() + m->indptr[r + 1];
        const auto it = std::lower_bound(lo, hi, c);
        if (it == hi || *it != c) return false;
        v = m->data[it - m->indices.begin()];
        return true;
    };
    for (int r = 0; r < N; ++r) {
        int slot_c = 0, slot_d = Ec_true;
        for (const auto &cv : rows[r]) {
            const int slot = cv.second ? slot_c++ : slot_d++;
            off[(size_t)slot * S + r] = cv.first * (int)sizeof(cplx);
            for (int o = 0; o < Lp1; ++o) {
                cplx v;
                if (value_at(ops[o], r, cv.first, v)) vals[((size_t)o * E + slot) * S + r] = v;
            }
        }
    }
    return true;
}

// csr_fw / csr_bw: [K*(1+L)] sparse operators and their conjugate transposes (pr->ops then holds their
// data arrays), or both NULL for dense row-major operators
static int engine_create(const kh_problem *pr, const kh_csr *csr_fw, const kh_csr *csr_bw, kh_engine **out) {
    if (pr == nullptr || out == nullptr) return kh_fail(KH_ERR_INVALID, "null argument");
    *out = nullptr;
    if (pr->K < 1 || pr->N < 1 || pr->L < 0 || pr->nt < 2)
        return kh_fail(KH_ERR_INVALID, "bad sizes K=%d N=%d L=%d nt=%d", pr->K, pr->N, pr->L, pr->nt);
    // (the register-resident families take up to KH_MAX_L controls; with more the generic kernels run, up to KH_GEN_MAX_L)
    if (pr->L > KH_GEN_MAX_L) return kh_fail(KH_ERR_UNSUPPORTED, "L=%d controls > %d", pr->L, KH_GEN_MAX_L);
    if (pr->dt == nullptr || pr->ops == nullptr) return kh_fail(KH_ERR_INVALID, "dt/ops missing");
    for (int n = 0; n < pr->nt - 1; ++n)
        if (!(pr->dt[n] > 0.0)) return kh_fail(KH_ERR_INVALID, "dt[%d] = %g is not positive", n, pr->dt[n]);
    const size_t nops = (size_t)pr->K * (1 + pr->L);
    for (int k = 0; k < pr->K; ++k)
        if (pr->ops[(size_t)k * (1 + pr->L)] == nullptr)
            return kh_fail(KH_ERR_INVALID, "objective %d has no drift operator", k);

    kh_engine *e = new kh_engine();
    e->K = pr->K;
    e->N = pr->N;
    e->L = pr->L;
    e->nt = pr->nt;
    e->is_super = pr->is_super ? 1 : 0;
    e->tol = pr->tol > 0.0 ? pr->tol : ldexp(1.0, -53);
    e->theta_max = pr->theta_max > 0.0 ? pr->theta_max : 1.0;
#define KH_HIP_E(call)                                                                   \
    do {                                                                                 \
        hipError_t _e = (call);                                                          \
        if (_e != hipSuccess) {                                                          \
            kh_engine_destroy(e);                                                        \
            return kh_fail(KH_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e));   \
        }                                                                                \
    } while (0)
    KH_HIP_E(hipGetDevice(&e->device));
    hipDeviceProp_t prop;
    KH_HIP_E(hipGetDeviceProperties(&prop, e->device));
    e->num_cus = prop.multiProcessorCount;
    // the generic kernels -- every engine's last resort -- keep four vectors of N elements in LDS: N <= 2540.  Sparse
    // operators up to N = 4096 may still run the streamed padded-row kernels (decided below: gen_fits stays false then
    // and whatever would need the generic kernels -- one launch per interval, more objectives than CUs -- is refused)
    const bool gen_fits = !(kh_gen_lds_bytes(e->N, csr_fw == nullptr) > (size_t)prop.sharedMemPerBlock && kh_gen_lds_bytes(e->N, csr_fw == nullptr) > 160 * 1024);
    e->gen_fits = gen_fits;
    if (!gen_fits && !(csr_fw != nullptr && e->N <= KH_ELLS_NMAX)) {
        kh_engine_destroy(e);
        return kh_fail(KH_ERR_UNSUPPORTED, "N=%d needs %zu bytes of LDS", pr->N, kh_gen_lds_bytes(pr->N, csr_fw == nullptr));
    }

    // ---- operator tables: forward pointers as given, adjoints staged once per distinct operator
    std::vector<const cplx *> fw(nops), bw(nops);
    std::map<const void *, cplx *> adj_of;
    for (size_t i = 0; i < nops; ++i) {
        const cplx *src = (const cplx *)pr->ops[i];
        fw[i] = src;
        if (src == nullptr) {
            bw[i] = nullptr;
            continue;
        }
        if (csr_fw != nullptr) {  // the caller supplies the conjugate transposes
            bw[i] = (const cplx *)csr_bw[i].data;
            continue;
        }
        auto it = adj_of.find(src);
        if (it == adj_of.end()) {
            cplx *dst = nullptr;
            KH_HIP_E(hipMalloc(&dst, sizeof(cplx) * (size_t)e->N * e->N));
            e->owned.push_back(dst);
            const int tiles = (e->N + 31) / 32;
            kh_adjoint_kernel<<<dim3(tiles, tiles), 256>>>(src, dst, e->N);
            it = adj_of.emplace(src, dst).first;
        }
        bw[i] = it->second;
    }
    KH_HIP_E(hipGetLastError());
    KH_HIP_E(hipMalloc((void **)&e->d_ops_fw, sizeof(cplx *) * nops));
    KH_HIP_E(hipMalloc((void **)&e->d_ops_bw, sizeof(cplx *) * nops));
    KH_HIP_E(hipMemcpy((void *)e->d_ops_fw, fw.data(), sizeof(cplx *) * nops, hipMemcpyHostToDevice));
    KH_HIP_E(hipMemcpy((void *)e->d_ops_bw, bw.data(), sizeof(cplx *) * nops, hipMemcpyHostToDevice));
    if (csr_fw == nullptr && e->L >= 1) {  // is every control operator its own (negative) adjoint, bit for bit?
        int *d_flags = nullptr, flags[3] = {1, 1, 1};
        KH_HIP_E(hipMalloc(&d_flags, sizeof(flags)));
        hipError_t err = hipMemset(d_flags, 0, sizeof(flags));
        if (err == hipSuccess) {
            kh_adjoint_sign_kernel<<<(unsigned)(nops < 1024 ? nops : 1024), 256>>>(e->d_ops_fw, e->d_ops_bw, (int)nops,
                                                                                   1 + e->L, e->N, d_flags);
            err = hipMemcpy(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost);
        }
        (void)hipFree(d_flags);
        KH_HIP_E(err);
        e->adj_sign = flags[0] == 0 ? 1.0 : (flags[1] == 0 ? -1.0 : 0.0);
        // every generator Hermitian and f = -+i: real spectrum (the q2 kernels' shorter series, kh_common.h)
        e->real_spectrum = flags[0] == 0 && flags[2] == 0 && !e->is_super;
        {   // f A anti-Hermitian up to a small Hermitian part of the drift (a weakly damped Liouvillian; a Hamiltonian
            // with a small anti-Hermitian part): the shorter series of kh_common.h applies with a margin
            const bool controls_ok = e->is_super ? flags[1] == 0 : flags[0] == 0;
            if (controls_ok) {
                unsigned long long *d_max = nullptr, bits = 0;
                KH_HIP_E(hipMalloc(&d_max, sizeof(bits)));
                hipError_t err2 = hipMemset(d_max, 0, sizeof(bits));
                if (err2 == hipSuccess) {
                    kh_herm_defect_kernel<<<(unsigned)(e->K < 1024 ? e->K : 1024), 256>>>(
                        e->d_ops_fw, e->d_ops_bw, (int)nops, 1 + e->L, e->N, e->is_super ? 1.0 : -1.0, d_max);
                    err2 = hipMemcpy(&bits, d_max, sizeof(bits), hipMemcpyDeviceToHost);
                }
                (void)hipFree(d_max);
                KH_HIP_E(err2);
                double fro2, dt_max = 0.0;
                memcpy(&fro2, &bits, sizeof(fro2));
                for (int n = 0; n < e->nt - 1; ++n) dt_max = pr->dt[n] > dt_max ? pr->dt[n] : dt_max;
                e->imag_defect = sqrt(fro2) * dt_max;
            }
        }
        if (const char *d = getenv("KH_TAYLOR"))  // A/B switch: plain Taylor coefficients everywhere
            if (atoi(d) != 0) {
                e->real_spectrum = false;
                e->imag_defect = -1.0;
            }
        if (const char *d = getenv("KH_NO_ADJ"))  // A/B switch: keep <chi|H phi> on the forward side
            if (atoi(d) != 0) e->adj_sign = 0.0;
    }
    KH_HIP_E(hipMalloc(&e->d_norms, sizeof(double) * nops));
    if (csr_fw != nullptr) {
        static_assert(sizeof(KhCsr) == sizeof(kh_csr), "kh_csr layout");
        KH_HIP_E(hipMalloc(&e->d_csr_fw, sizeof(KhCsr) * nops));
        KH_HIP_E(hipMalloc(&e->d_csr_bw, sizeof(KhCsr) * nops));
        KH_HIP_E(hipMemcpy(e->d_csr_fw, csr_fw, sizeof(KhCsr) * nops, hipMemcpyHostToDevice));
        KH_HIP_E(hipMemcpy(e->d_csr_bw, csr_bw, sizeof(KhCsr) * nops, hipMemcpyHostToDevice));
    }
    bool ell_ok = false;
    if (csr_fw != nullptr) {
        // canonical host copies of every distinct operator and of its conjugate transpose (small: a few entries per row)
        std::map<const void *, HostCsr> host_fw, host_bw;
        for (size_t i = 0; i < nops; ++i) {
            if (fw[i] == nullptr || host_fw.count(fw[i])) continue;
            KH_HIP_E(fetch_csr(csr_fw[i], e->N, host_fw[fw[i]]));
            KH_HIP_E(fetch_csr(csr_bw[i], e->N, host_bw[fw[i]]));
        }
        // what the dense engines ask the device (kh_adjoint_sign_kernel, kh_herm_defect_kernel): is every generator
        // Hermitian (real spectrum), or anti-Hermitian up to a small part of the drift?  -> the shorter series
        if (e->L >= 1) {
            bool ctl_plus = true, ctl_minus = true, drift_plus = true;
            double fro2 = 0.0;
            for (size_t i = 0; i < nops; ++i) {
                if (fw[i] == nullptr) continue;
                const HostCsr &a = host_fw[fw[i]], &b = host_bw[fw[i]];
                if (i % (size_t)(1 + e->L) == 0) {
                    drift_plus = drift_plus && csr_equal(a, b, 1.0);
                    fro2 = std::max(fro2, csr_part_fro2(a, b, e->is_super ? 1.0 : -1.0));
                } else {
                    ctl_plus = ctl_plus && csr_equal(a, b, 1.0);
                    ctl_minus = ctl_minus && csr_equal(a, b, -1.0);
                }
            }
            e->adj_sign = ctl_plus ? 1.0 : (ctl_minus ? -1.0 : 0.0);
            e->real_spectrum = ctl_plus && drift_plus && !e->is_super;
            if (e->is_super ? ctl_minus : ctl_plus) {
                double dt_max = 0.0;
                for (int n = 0; n < e->nt - 1; ++n) dt_max = pr->dt[n] > dt_max ? pr->dt[n] : dt_max;
                e->imag_defect = sqrt(fro2) * dt_max;
            }
            if (const char *d = getenv("KH_TAYLOR"))
                if (atoi(d) != 0) {
                    e->real_spectrum = false;
                    e->imag_defect = -1.0;
                }
        }
        // the padded row form (kh_ell.h): one structure per distinct operator list and direction
        const char *force_k = getenv("KH_KERNEL");
        // ... matrix in registers where the rows fit (N <= 2048), else -- or with KH_KERNEL=ellstream -- the streamed form
        // (N <= 4096, rows up to 32 entries): the same pools with their own row count, read per term
        const bool want_stream = force_k && strcmp(force_k, "ellstream") == 0;
        for (int form = want_stream ? 1 : 0; form < 2 && !ell_ok; ++form) {
            const bool stream = form == 1;
            if (e->N > (stream ? KH_ELLS_NMAX : KH_ELL_NMAX) || e->L > KH_MAX_L || (force_k && strcmp(force_k, "generic") == 0)) continue;
            if (stream && getenv("KH_NO_ELLSTREAM") && atoi(getenv("KH_NO_ELLSTREAM"))) continue;
            ell_ok = true;
            e->ell_E = 0;
            int ec_max = 0;
            std::map<std::vector<const void *>, std::pair<KhEll, KhEll>> made;
            std::vector<KhEll> ell_fw(e->K), ell_bw(e->K);
            std::vector<int> off_pool;
            std::vector<cplx> vals_pool;
            for (int k = 0; k < e->K && ell_ok; ++k) {
                std::vector<const void *> key(fw.begin() + (size_t)k * (1 + e->L), fw.begin() + (size_t)(k + 1) * (1 + e->L));
                auto it = made.find(key);
                if (it == made.end()) {
                    KhEll pair[2];
                    for (int dir = 0; dir < 2 && ell_ok; ++dir) {
                        std::vector<const HostCsr *> ops_h;
                        for (const void *ptr : key)
                            ops_h.push_back(ptr == nullptr ? nullptr : (dir == 0 ? &host_fw[ptr] : &host_bw[ptr]));
                        std::vector<int> off;
                        std::vector<cplx> vals;
                        int E = 0, Ec = 0;
                        if (!build_ell_host(ops_h, e->N, off, vals, E, Ec, stream)) {
                            ell_ok = false;
                            break;
                        }
                        ec_max = std::max(ec_max, Ec);
                        pair[dir].off_at = (long long)off_pool.size();
                        pair[dir].vals_at = (long long)vals_pool.size();
                        off_pool.insert(off_pool.end(), off.begin(), off.end());
                        vals_pool.insert(vals_pool.end(), vals.begin(), vals.end());
                        pair[dir].E = E;
                        pair[dir].Ec = Ec;
                        pair[dir].rows = stream ? (e->N + 63) / 64 * 64 : kh_ell_rows(e->N);
                        pair[dir].pad_ = 0;
                        e->ell_E = std::max(e->ell_E, E);
                    }
                    if (!ell_ok) break;
                    it = made.emplace(key, std::make_pair(pair[0], pair[1])).first;
                }
                ell_fw[k] = it->second.first;
                ell_bw[k] = it->second.second;
            }
            if (ell_ok) {
                KH_HIP_E(hipMalloc(&e->d_ell_off, sizeof(int) * off_pool.size()));
                KH_HIP_E(hipMalloc(&e->d_ell_vals, sizeof(cplx) * vals_pool.size()));
                KH_HIP_E(hipMemcpy(e->d_ell_off, off_pool.data(), sizeof(int) * off_pool.size(), hipMemcpyHostToDevice));
                KH_HIP_E(hipMemcpy(e->d_ell_vals, vals_pool.data(), sizeof(cplx) * vals_pool.size(), hipMemcpyHostToDevice));
                KH_HIP_E(hipMalloc(&e->d_ell_fw, sizeof(KhEll) * e->K));
                KH_HIP_E(hipMalloc(&e->d_ell_bw, sizeof(KhEll) * e->K));
                KH_HIP_E(hipMemcpy(e->d_ell_fw, ell_fw.data(), sizeof(KhEll) * e->K, hipMemcpyHostToDevice));
                KH_HIP_E(hipMemcpy(e->d_ell_bw, ell_bw.data(), sizeof(KhEll) * e->K, hipMemcpyHostToDevice));
                e->ell_stream = stream;
                if (stream) {
                    // one scratch plane per workgroup (update sweep: K of them; plain sweeps: at most one per CU)
                    e->ell_scratch_stride = (long long)std::max(ec_max, 4) * ((e->N + 63) / 64 * 64);
                    const int wgs = e->K < e->num_cus ? e->K : e->num_cus;  // (the update sweep takes K <= #CUs workgroups, the plain sweeps at most #CUs)
                    KH_HIP_E(hipMalloc(&e->d_ell_scratch, sizeof(cplx) * (size_t)e->ell_scratch_stride * wgs));
                }
            }
        }
    }
    if (!e->gen_fits && !ell_ok) {
        kh_engine_destroy(e);
        return kh_fail(KH_ERR_UNSUPPORTED, "N=%d: rows wider than 32 entries and no room for the generic kernels' vectors in LDS", pr->N);
    }
    if (pr->op_norms != nullptr) {
        KH_HIP_E(hipMemcpy(e->d_norms, pr->op_norms, sizeof(double) * nops, hipMemcpyHostToDevice));
    } else {
        kh_fro_norms<<<(unsigned)nops, 256>>>(e->d_ops_fw, (int)nops, e->N, e->d_norms);
        KH_HIP_E(hipGetLastError());
    }
    {
        double tab[KH_MAX_DEGREE + 1];
        kh_build_degree_table(e->tol, tab);
        KH_HIP_E(hipMalloc(&e->d_deg_theta, sizeof(tab)));
        KH_HIP_E(hipMemcpy(e->d_deg_theta, tab, sizeof(tab), hipMemcpyHostToDevice));
    }
    KH_HIP_E(hipMalloc(&e->d_dt, sizeof(double) * (e->nt - 1)));
    KH_HIP_E(hipMemcpy(e->d_dt, pr->dt, sizeof(double) * (e->nt - 1), hipMemcpyHostToDevice));

    // ---- kernel family
    e->kind = KIND_GENERIC;
    const int max_wgs = e->num_cus < 64 * KH_GATHER_CHUNKS ? e->num_cus : 64 * KH_GATHER_CHUNKS;
    e->grid_update = e->K < max_wgs ? e->K : max_wgs;
    if (const char *d = getenv("KH_COOP_LAUNCH")) e->coop_launch = atoi(d) != 0;
    if (const char *d = getenv("KH_Q2_SINGLE")) e->q2_single = atoi(d) != 0;
    if (const char *d = getenv("KH_TILE_SINGLE")) e->tile_single = atoi(d) != 0;
    if (const char *d = getenv("KH_COOP_SINGLE")) e->coop_single = atoi(d) != 0;
    if (const char *d = getenv("KH_P2P_FAIL_AT")) e->p2p_fail_at = atoi(d);
    if (const char *d = getenv("KH_P2P_FAIL_RANK")) e->p2p_fail_rank = atoi(d);
    if (const char *d = getenv("KH_P2P_FAIL_SWEEP")) e->p2p_fail_sweep = atoi(d);
    if (const char *d = getenv("KH_Q4")) e->use_q4 = atoi(d) != 0;
    if (const char *d = getenv("KH_POLL_DELAY")) {
        e->poll_delay = atoi(d);
        e->poll_delay_set = true;
    }
    if (const char *d = getenv("KH_ADJ_DELAY")) e->adj_poll_delay = atoi(d);
    if (const char *d = getenv("KH_COOP_DELAY")) e->coop_poll_delay = atoi(d);
    if (const char *d = getenv("KH_TIMEOUT_MS"))  // e.g. under a profiler that slows the kernels down
        if (atoll(d) > 0) {
            e->timeout_ticks = atoll(d) * 100000LL;
            e->timeout_set = true;
        }
    const char *force = getenv("KH_KERNEL");  // "generic" | "tile256" | "tile512" | "q2" | "coop" (testing)
    const bool tile_ok = csr_fw == nullptr && e->N <= KH_TILE_N && e->L >= 1 && e->L <= 4 && e->K <= max_wgs;
    // More objectives than CUs, one control: 256-thread workgroups (one wave per SIMD, 256 VGPRs) fit two per
    // CU, so up to 2 x #CUs objectives stay co-resident -- and the two workgroups of a CU hide each other's
    // phase latency.
    const int max_wgs2 = 2 * e->num_cus < 64 * KH_GATHER_CHUNKS_WIDE ? 2 * e->num_cus : 64 * KH_GATHER_CHUNKS_WIDE;
    const bool tile2_ok = csr_fw == nullptr && e->N <= KH_TILE_N && e->L == 1 && e->K > max_wgs && e->K <= max_wgs2;
    if (tile2_ok && !(force && strcmp(force, "generic") == 0)) {
        e->kind = KIND_TILE_RPT2;
        e->grid_update = e->K;
    }
    // More objectives than can be co-resident (so no in-kernel exchange), tile-sized: the register-tile kernel with
    // ONE LAUNCH PER INTERVAL (the form the sharded sweep uses, kh_update_step) -- every launch re-stages the two
    // operator tiles of its objectives (128 KiB each, from L2 / the Infinity Cache), which still beats the generic
    // kernels' re-streaming of the operators for every term by 5x (K = 1024: 207 -> see DESIGN.md us per interval).
    const bool tile_step_ok = csr_fw == nullptr && e->N <= KH_TILE_N && e->L >= 1 && e->L <= 4 && !tile_ok && !tile2_ok &&
                              e->K > max_wgs && !(getenv("KH_NO_STEPWISE") && atoi(getenv("KH_NO_STEPWISE")));
    if (tile_step_ok && force == nullptr) {
        e->kind = KIND_TILE_RPT1;
        e->grid_update = e->K;
        e->stepwise_only = true;
        // one workgroup per CU; as few workgroups as give everybody the same number of objectives (K = 384, two controls:
        // 192 x 2 in 23.9 us per interval against 128 x 2 + 128 x 1 in 24.5)
        int G = e->num_cus < 64 * KH_GATHER_CHUNKS ? e->num_cus : 64 * KH_GATHER_CHUNKS;
        if (e->K > G) G = (e->K + (e->K + G - 1) / G - 1) / ((e->K + G - 1) / G);
        if (const char *g = getenv("KH_STREAM_G"))  // testing
            if (atoi(g) >= 1 && atoi(g) <= G) G = atoi(g);
        if (G > e->K) G = e->K;
        e->stream_G = G;
        e->stream = (long long)G * KH_STREAM_MMAX >= e->K && !(getenv("KH_NO_STREAM") && atoi(getenv("KH_NO_STREAM")));
    }
    if (tile_ok && !(force && strcmp(force, "generic") == 0)) {
        // two waves per SIMD are needed to keep the fp64 FMA pipe issuing back to back
        e->kind = KIND_TILE_RPT1;
        if (e->L == 1) e->kind = KIND_TILE_Q2;  // two Taylor terms per phase (kh_tile64q2.h)
        if (force && strcmp(force, "tile512") == 0) e->kind = KIND_TILE_RPT1;
        if (force && strcmp(force, "tile256") == 0 && e->L == 1) e->kind = KIND_TILE_RPT2;  // (two controls: 204 spilled values, never a default choice -- no such instantiation any more)
        e->grid_update = e->K;
        // small problems: one wave per objective, the objectives of the GPU in one workgroup (kh_mini.h);
        // KH_KERNEL=q2 keeps the workgroup-per-objective kernels, KH_KERNEL=mini is accepted for symmetry
        e->mini = e->kind == KIND_TILE_Q2 && e->N <= KH_MINI_N && e->K <= KH_MINI_MAXK && force == nullptr;
    }
    if (force && strcmp(force, "mini") == 0 && tile_ok && e->L == 1 && e->N <= KH_MINI_N && e->K <= KH_MINI_MAXK) {
        e->kind = KIND_TILE_Q2;
        e->grid_update = e->K;
        e->mini = true;
    }
    e->quad = e->mini && e->N <= KH_QUAD_N && e->K <= KH_QUAD_MAXK && !(force && strcmp(force, "mini") == 0);
    // objectives sharing ONE operator list with a state too large for a register tile: one Taylor
    // term of all objectives is a dense (N x N)(N x K) product -> fp64 matrix cores (kh_coop.h)
    {
        bool shared = true;
        for (size_t i = 0; i < nops && shared; ++i) shared = fw[i] == fw[i % (size_t)(1 + e->L)];
        // objectives per workgroup: as few as keeps the grid within the co-resident limit (a round is bound by
        // the block fetch, which shrinks with the column count; the MFMA work per workgroup does not grow)
        const int G = (e->N + 15) / 16;
        int cols = G * ((e->K + 3) / 4) <= max_wgs ? 4 : KH_COOP_COLS;
        // two objectives per workgroup (half the matrix-core work of a round per workgroup, twice the workgroups)
        // where every column group still gets an XCD of its own (kh_coop_place): up to 8 groups of at most 32
        if (e->K > 8 && (e->K + 1) / 2 <= 8 && G <= 32 && G * ((e->K + 1) / 2) <= max_wgs) cols = 2;
        if (const char *cenv = getenv("KH_COOP_COLS")) {  // testing
            const int want = atoi(cenv);
            if ((want == 2 || want == 4 || want == 16) && G * ((e->K + want - 1) / want) <= max_wgs)
                cols = want;
        }
        const int Y = (e->K + cols - 1) / cols;
        const bool forced = force && strcmp(force, "coop") == 0;
        const bool fits = csr_fw == nullptr && shared && e->N <= 480 && e->L <= KH_COOP_MAX_L && G * Y <= max_wgs;
        if (fits && (forced || (e->N > KH_TILE_N && force == nullptr))) {
            e->kind = KIND_COOP;
            e->coop_G = G;
            e->coop_Y = Y;
            e->coop_cols = cols;
            e->coop_ks = cols <= 4 ? kh_coop4_slots(e->N) : (e->N + 31) / 32;  // operator-fragment slots per lane
            e->coop_vbuf_bytes = sizeof(kh_u64) * KH_COOP_RING * (size_t)Y * G * 16 * KH_COOP_COLS * 4;
            KH_HIP_E(hipMalloc(&e->d_coop_vbuf, e->coop_vbuf_bytes));
            KH_HIP_E(hipMalloc(&e->d_coop_xcc, sizeof(unsigned int) * (size_t)G * Y));
            e->coop_xcd = cols <= 4 && Y <= 8 && G <= 32 && !(getenv("KH_COOP_XCD") && atoi(getenv("KH_COOP_XCD")) == 0);
            e->grid_update = e->K < max_wgs ? e->K : max_wgs;  // (stepwise launches use the generic kernel)
            // A round (one Taylor term) costs a cross-workgroup exchange here, so fewer, longer
            // sub-steps pay: theta <= 4 needs ~31 terms per sub-step against 4 x 18 at theta <= 1.
            // Round-off grows like e^theta (55 eps per step at theta = 4), still far inside the
            // parity budget (measured: unchanged 3e-15 vs the oracle on the transmon Liouvillians).
            if (!(pr->theta_max > 0.0)) e->theta_max = 4.0;
        }
    }
    // Per-objective operators with 64 < N <= 128: the generator in registers (kh_tilen.h) instead of the generic kernels'
    // re-streaming of every operator for every term.  (Objectives sharing one operator list took the cooperative
    // matrix-core kernels above; KH_KERNEL=tilen forces this family for them too: testing.)
    bool tilen_ok = false;
    {
        const bool forced = force && strcmp(force, "tilen") == 0;
        if (csr_fw == nullptr && e->N > KH_TILE_N && e->N <= KH_TN_NMAX && e->L >= 1 && e->L <= KH_MAX_L &&
            ((e->kind == KIND_GENERIC && force == nullptr) || forced)) {
            tilen_ok = true;
            if (forced) {
                e->kind = KIND_GENERIC;  // (undo the cooperative choice)
                e->grid_update = e->K < max_wgs ? e->K : max_wgs;
                if (!(pr->theta_max > 0.0)) e->theta_max = 1.0;
            }
            std::map<const void *, cplx *> perm_of;
            for (int dir = 0; dir < 2; ++dir) {
                const std::vector<const cplx *> &tab = dir == 0 ? fw : bw;
                std::vector<const cplx *> out(nops, nullptr);
                for (size_t i = 0; i < nops; ++i) {
                    if (tab[i] == nullptr) continue;
                    auto it = perm_of.find(tab[i]);
                    if (it == perm_of.end()) {
                        cplx *dst = nullptr;
                        KH_HIP_E(hipMalloc(&dst, sizeof(cplx) * (KH_TN_NMAX / 4) * KH_TN_THREADS));
                        e->owned.push_back(dst);
                        kh_tn_permute<<<KH_TN_NMAX / 4, KH_TN_THREADS>>>(tab[i], dst, e->N);
                        it = perm_of.emplace(tab[i], dst).first;
                    }
                    out[i] = it->second;
                }
                const cplx ***slot = dir == 0 ? &e->d_tn_fw : &e->d_tn_bw;
                KH_HIP_E(hipMalloc((void **)slot, sizeof(cplx *) * nops));
                KH_HIP_E(hipMemcpy((void *)*slot, out.data(), sizeof(cplx *) * nops, hipMemcpyHostToDevice));
            }
            KH_HIP_E(hipGetLastError());
            e->tn_h1reg = e->L == 1 && e->N <= 96 && !(getenv("KH_TN_H1REG") && atoi(getenv("KH_TN_H1REG")) == 0);
            for (int k = 0; k < e->K; ++k) e->tn_h1reg = e->tn_h1reg && fw[(size_t)k * 2 + 1] != nullptr;
            if (e->K <= max_wgs) {
                e->kind = KIND_TILEN;
                e->grid_update = e->K;
            }
        }
    }
    // Five to eight controls, N <= 64: the register-tile kernels with the operators beyond the CU's room streamed
    // (kh_tile64x.h) instead of the generic kernels.  The plain sweeps take their objectives in turns (any K); the
    // update sweep needs one resident workgroup per objective, first order and the adjoint-side store (launch_update:
    // otherwise the generic kernels, which stay this engine's `kind`).  KH_TX=0: off (A/B switch); KH_KERNEL=tilex: testing
    bool tx_ok = false;
    {
        const bool forced = force && strcmp(force, "tilex") == 0;
        const bool off = getenv("KH_TX") && atoi(getenv("KH_TX")) == 0;
        if (csr_fw == nullptr && e->N <= KH_TILE_N && e->L >= KH_TX_MIN_L && e->L <= KH_MAX_L && e->kind == KIND_GENERIC &&
            (force == nullptr || forced) && !off) {
            tx_ok = true;
            std::map<const void *, cplx *> perm_of;
            cplx *zero_tile = nullptr;  // stands in for a control an objective does not have (no branches in the kernels' loads)
            KH_HIP_E(hipMalloc(&zero_tile, sizeof(cplx) * 8 * KH_TX_THREADS));
            e->owned.push_back(zero_tile);
            KH_HIP_E(hipMemset(zero_tile, 0, sizeof(cplx) * 8 * KH_TX_THREADS));
            for (int dir = 0; dir < 2; ++dir) {
                const std::vector<const cplx *> &tab = dir == 0 ? fw : bw;
                std::vector<const cplx *> out(nops, zero_tile);
                for (size_t i = 0; i < nops; ++i) {
                    if (tab[i] == nullptr) continue;
                    auto it = perm_of.find(tab[i]);
                    if (it == perm_of.end()) {
                        cplx *dst = nullptr;
                        KH_HIP_E(hipMalloc(&dst, sizeof(cplx) * 8 * KH_TX_THREADS));
                        e->owned.push_back(dst);
                        kh_tx_permute<<<8, KH_TX_THREADS>>>(tab[i], dst, e->N);
                        it = perm_of.emplace(tab[i], dst).first;
                    }
                    out[i] = it->second;
                }
                const cplx ***slot = dir == 0 ? &e->d_tx_fw : &e->d_tx_bw;
                KH_HIP_E(hipMalloc((void **)slot, sizeof(cplx *) * nops));
                KH_HIP_E(hipMemcpy((void *)*slot, out.data(), sizeof(cplx *) * nops, hipMemcpyHostToDevice));
            }
            KH_HIP_E(hipGetLastError());
            e->tx_update = e->K <= max_wgs;
        }
    }
    // Sparse operators in the padded row form: one 1024-thread workgroup per objective, the matrix in registers
    // (kh_ell.h).  The update sweep exchanges the sums in-kernel, so all K workgroups must be resident (one per CU);
    // with more objectives it stays with the generic CSR kernels, the plain sweeps take their objectives in turns.
    if (ell_ok) {
        if (e->K <= max_wgs) {
            e->kind = KIND_ELL;
            e->grid_update = e->K;
        }
        // a term of the series costs a workgroup-wide round whatever it multiplies: fewer, longer sub-steps pay, as
        // for the cooperative kernels (theta <= 4: round-off ~ e^theta eps per step, far inside the parity budget)
        if (!(pr->theta_max > 0.0)) {
            // (the long sub-steps only with the Chebyshev form's coefficients; Taylor's at theta <= 4 as before)
            const bool cheb = (e->real_spectrum || (e->imag_defect >= 0.0 && e->imag_defect <= 0.05)) &&
                              !(getenv("KH_NEAR_IMAG") && atoi(getenv("KH_NEAR_IMAG")) == 0);
            e->theta_max = cheb ? KH_ELL_THETA_CAP : 4.0;
            // KH_ELL_CAP (scripts/exp_ell_cap.py): another cap for the Chebyshev-form tables only, never beyond the one
            // they were validated for -- plain Taylor keeps theta <= 4 (its round-off grows like e^theta)
            if (const char *d = getenv("KH_ELL_CAP"))
                if (cheb && atof(d) > 0.0) e->theta_max = atof(d) < KH_ELL_THETA_CAP ? atof(d) : KH_ELL_THETA_CAP;
        }
    }
    // The plain sweeps have no cross-objective coupling, so the register-tile kernel serves them for any
    // number of objectives (workgroups simply run in turns) even when the update sweep needs the generic one.
    e->kind_store = ell_ok ? KIND_ELL : (tilen_ok ? KIND_TILEN : (tx_ok ? KIND_TILEX : e->kind));
    if (e->kind == KIND_GENERIC && csr_fw == nullptr && e->N <= KH_TILE_N && e->L >= 1 && e->L <= 4 &&
        !(force && strcmp(force, "generic") == 0))
        e->kind_store = KIND_TILE_RPT1;
    // ... and with one control that kernel is the two-terms-per-phase one (kh_q2_sweep_store takes its objectives in
    // turns: no co-residency needed), whatever the update sweep has to use: K = 512 on one GPU 7.5 -> 5.8 us per
    // interval of the backward sweep (two turns of the 256-objective sweep instead of the one-term-per-phase kernel
    // with two workgroups per CU).  KH_Q2_STORE=0: the update sweep's own family (A/B switch)
    if ((e->kind_store == KIND_TILE_RPT2 || e->kind_store == KIND_TILE_RPT1) && e->L == 1 && force == nullptr &&
        !(getenv("KH_Q2_STORE") && atoi(getenv("KH_Q2_STORE")) == 0))
        e->kind_store = KIND_TILE_Q2;
    // (not for 16 operator slots per lane x 16 objectives per workgroup -- N > 256 with more objectives than 4 per
    // workgroup keep co-resident --: the A^2 chain's second fragment does not fit the registers there, launch_coop_store)
    const bool coop_sq = e->kind == KIND_COOP && e->L == 1 && !(getenv("KH_COOP_NOSQ") && atoi(getenv("KH_COOP_NOSQ"))) &&
                         !(e->coop_cols == 16 && e->coop_ks > 8);
    if (e->kind == KIND_TILE_Q2 || e->kind_store == KIND_TILE_Q2 || coop_sq) {
        // stage P0 = H0 H0, P1 = H0 H1 + H1 H0, P2 = H1 H1 once per distinct operator (pair)
        const unsigned pgrid = (unsigned)(((size_t)e->N * e->N + 255) / 256 < 16 ? 16 : ((size_t)e->N * e->N + 255) / 256);
        const size_t bytes = sizeof(cplx) * (size_t)e->N * e->N;
        for (int dir = 0; dir < 2; ++dir) {
            const std::vector<const cplx *> &tab = dir == 0 ? fw : bw;
            std::vector<const cplx *> sq((size_t)e->K * 3, nullptr);
            std::map<const void *, cplx *> p0_of, p2_of;
            std::map<std::pair<const void *, const void *>, cplx *> p1_of;
            for (int k = 0; k < e->K; ++k) {
                const cplx *H0 = tab[(size_t)k * 2], *H1 = tab[(size_t)k * 2 + 1];
                auto it0 = p0_of.find(H0);
                if (it0 == p0_of.end()) {
                    cplx *dst = nullptr;
                    KH_HIP_E(hipMalloc(&dst, bytes));
                    e->owned.push_back(dst);
                    kh_q2_product<<<pgrid, 256>>>(H0, H0, dst, e->N, 0);
                    it0 = p0_of.emplace(H0, dst).first;
                }
                sq[(size_t)k * 3] = it0->second;
                if (H1 == nullptr) continue;
                auto it2 = p2_of.find(H1);
                if (it2 == p2_of.end()) {
                    cplx *dst = nullptr;
                    KH_HIP_E(hipMalloc(&dst, bytes));
                    e->owned.push_back(dst);
                    kh_q2_product<<<pgrid, 256>>>(H1, H1, dst, e->N, 0);
                    it2 = p2_of.emplace(H1, dst).first;
                }
                sq[(size_t)k * 3 + 2] = it2->second;
                auto key = std::make_pair((const void *)H0, (const void *)H1);
                auto it1 = p1_of.find(key);
                if (it1 == p1_of.end()) {
                    cplx *dst = nullptr;
                    KH_HIP_E(hipMalloc(&dst, bytes));
                    e->owned.push_back(dst);
                    kh_q2_product<<<pgrid, 256>>>(H0, H1, dst, e->N, 1);
                    it1 = p1_of.emplace(key, dst).first;
                }
                sq[(size_t)k * 3 + 1] = it1->second;
            }
            KH_HIP_E(hipGetLastError());
            const cplx ***slot = dir == 0 ? &e->d_sq_fw : &e->d_sq_bw;
            KH_HIP_E(hipMalloc((void **)slot, sizeof(cplx *) * sq.size()));
            KH_HIP_E(hipMemcpy((void *)*slot, sq.data(), sizeof(cplx *) * sq.size(), hipMemcpyHostToDevice));
        }
    }
    if (e->kind == KIND_COOP) {
        // fragment-ordered copies of the (shared) operators and, for one control, of P0, P1, P2 (kh_coop.h)
        const size_t elems = kh_coop_table_elems(e->coop_G, e->coop_ks);  // (row blocks padded apart: kh_coop_table_stride)
        const size_t frag_elems = (size_t)e->coop_G * KH_COOP_WAVES * e->coop_ks * 64;
        std::map<const void *, const cplx *> perm_of;
        auto permuted = [&](const cplx *src, const cplx **out) -> hipError_t {
            *out = nullptr;
            if (src == nullptr) return hipSuccess;
            auto it = perm_of.find(src);
            if (it == perm_of.end()) {
                cplx *dst = nullptr;
                // (+ one zero-slot word per (row block, wave) behind the table: kh_coop_mask_kernel)
                const hipError_t err = hipMalloc(&dst, sizeof(cplx) * elems + sizeof(unsigned int) * e->coop_G * KH_COOP_WAVES);
                if (err != hipSuccess) return err;
                e->owned.push_back(dst);
                {
                    const hipError_t merr = hipMemset(dst, 0, sizeof(cplx) * elems);
                    if (merr != hipSuccess) return merr;
                }
                kh_coop_permute_kernel<<<(unsigned)((frag_elems + 255) / 256), 256>>>(src, dst, e->N, e->coop_G, e->coop_ks, e->coop_cols);
                kh_coop_mask_kernel<<<e->coop_G * KH_COOP_WAVES, 64>>>(dst, (unsigned int *)(dst + elems), e->coop_ks);
                it = perm_of.emplace(src, dst).first;
            }
            *out = it->second;
            return hipSuccess;
        };
        for (int dir = 0; dir < 2; ++dir) {
            const std::vector<const cplx *> &tab = dir == 0 ? fw : bw;
            std::vector<const cplx *> fops(1 + e->L, nullptr), sq3(3, nullptr);
            for (int o = 0; o <= e->L; ++o) KH_HIP_E(permuted(tab[o], &fops[o]));
            const cplx ***slot = dir == 0 ? &e->d_coop_fops_fw : &e->d_coop_fops_bw;
            KH_HIP_E(hipMalloc((void **)slot, sizeof(cplx *) * fops.size()));
            KH_HIP_E(hipMemcpy((void *)*slot, fops.data(), sizeof(cplx *) * fops.size(), hipMemcpyHostToDevice));
            if (coop_sq) {
                std::vector<const cplx *> nat(3, nullptr);
                KH_HIP_E(hipMemcpy(nat.data(), dir == 0 ? e->d_sq_fw : e->d_sq_bw, sizeof(cplx *) * 3, hipMemcpyDeviceToHost));
                for (int i = 0; i < 3; ++i) KH_HIP_E(permuted(nat[i], &sq3[i]));
                const cplx ***sslot = dir == 0 ? &e->d_coop_sq_fw : &e->d_coop_sq_bw;
                KH_HIP_E(hipMalloc((void **)sslot, sizeof(cplx *) * 3));
                KH_HIP_E(hipMemcpy((void *)*sslot, sq3.data(), sizeof(cplx *) * 3, hipMemcpyHostToDevice));
            }
        }
        KH_HIP_E(hipGetLastError());
#ifdef KH_WITH_C4W
        if (coop_sq && e->coop_cols == 2 && e->L == 1 && bw[1] != nullptr && kh_c4_groups(e->N) <= 30 &&
            getenv("KH_COOP4W") != nullptr && atoi(getenv("KH_COOP4W")) != 0) {
            // the same five tables in the fragment order of kh_coop4w.h
            const int NG = kh_c4_groups(e->N);
            const size_t elems = kh_c4_table_elems(e->coop_G, NG), frag = (size_t)e->coop_G * KH_C4_WAVES * NG * 64;
            std::map<const void *, const cplx *> perm4;
            auto permuted4 = [&](const cplx *src, const cplx **out) -> hipError_t {
                *out = nullptr;
                if (src == nullptr) return hipSuccess;
                auto it = perm4.find(src);
                if (it == perm4.end()) {
                    cplx *dst = nullptr;
                    const hipError_t err = hipMalloc(&dst, sizeof(cplx) * elems + sizeof(unsigned int) * e->coop_G * KH_C4_WAVES);
                    if (err != hipSuccess) return err;
                    e->owned.push_back(dst);
                    const hipError_t merr = hipMemset(dst, 0, sizeof(cplx) * elems);
                    if (merr != hipSuccess) return merr;
                    kh_c4_permute_kernel<<<(unsigned)((frag + 255) / 256), 256>>>(src, dst, e->N, e->coop_G, NG);
                    kh_c4_mask_kernel<<<e->coop_G * KH_C4_WAVES, 64>>>(dst, (unsigned int *)(dst + elems), NG);
                    it = perm4.emplace(src, dst).first;
                }
                *out = it->second;
                return hipSuccess;
            };
            for (int dir = 0; dir < 2; ++dir) {
                const std::vector<const cplx *> &tab = dir == 0 ? fw : bw;
                std::vector<const cplx *> fops(2, nullptr), sq3(3, nullptr), nat(3, nullptr);
                for (int o = 0; o < 2; ++o) KH_HIP_E(permuted4(tab[o], &fops[o]));
                KH_HIP_E(hipMemcpy(nat.data(), dir == 0 ? e->d_sq_fw : e->d_sq_bw, sizeof(cplx *) * 3, hipMemcpyDeviceToHost));
                for (int i = 0; i < 3; ++i) KH_HIP_E(permuted4(nat[i], &sq3[i]));
                const cplx ***fslot = dir == 0 ? &e->d_c4_fops_fw : &e->d_c4_fops_bw;
                const cplx ***sslot = dir == 0 ? &e->d_c4_sq_fw : &e->d_c4_sq_bw;
                KH_HIP_E(hipMalloc((void **)fslot, sizeof(cplx *) * 2));
                KH_HIP_E(hipMemcpy((void *)*fslot, fops.data(), sizeof(cplx *) * 2, hipMemcpyHostToDevice));
                KH_HIP_E(hipMalloc((void **)sslot, sizeof(cplx *) * 3));
                KH_HIP_E(hipMemcpy((void *)*sslot, sq3.data(), sizeof(cplx *) * 3, hipMemcpyHostToDevice));
            }
            KH_HIP_E(hipGetLastError());
            e->coop4w = true;
            e->c4_NG = NG;
        }
#endif
        if (coop_sq && bw[1] != nullptr && !(getenv("KH_COOP_NO_ADJ") && atoi(getenv("KH_COOP_NO_ADJ")))) {
            e->coop_adj = true;
            e->coop_adj_op = bw[1];
            KH_HIP_E(hipMalloc(&e->d_coop_adj_nz, (size_t)e->coop_G * e->coop_G));
            kh_coop_adj_mask_kernel<<<e->coop_G * e->coop_G, 256>>>(bw[1], e->N, e->coop_G, e->d_coop_adj_nz);
            KH_HIP_E(hipGetLastError());
        }
    }
    {  // (every kernel family but the cooperative one reads the series tables)
        std::vector<double> tab(KH_MAX_DEGREE + 1), c0(KH_MAX_DEGREE + 1), rows((size_t)(KH_MAX_DEGREE + 1) * KH_Q2_ROWS * 2),
            ratios((size_t)(KH_MAX_DEGREE + 1) * KH_RATIO_STRIDE);
        // the cooperative kernels (A^2 chain, one control): a term is a cross-workgroup round, so the Chebyshev form
        // is used up to theta = 4 and also for generators that are anti-Hermitian only up to a small defect
        e->coop_series = e->kind == KIND_COOP && coop_sq && e->imag_defect >= 0.0 && e->imag_defect <= 0.05;
        if (e->coop_series) {
            kh_build_real_spectrum_rows(e->tol, tab.data(), c0.data(), rows.data(), ratios.data(), 4.0, e->imag_defect);
        } else if (ell_ok && (e->real_spectrum || (e->imag_defect >= 0.0 && e->imag_defect <= 0.05)) &&
                   !(getenv("KH_NEAR_IMAG") && atoi(getenv("KH_NEAR_IMAG")) == 0)) {
            // sparse operators in the padded row form: a term costs a workgroup-wide round, so one long sub-step beats
            // several short ones: the Chebyshev form up to theta = 6.  Measured on the reference's three-states problem
            // (scripts/exp_ell_cap.py, theta = 4.4 ... 7.9 per step, 3 iterations x 3 sweeps x 2000 steps): tau within
            // 6e-14 and the pulses within 7e-15 of the same run with theta <= 1 per sub-step, for caps 4, 5, 6 and 8
            // alike; KH_ELL_CAP: A/B switch
            double cap = KH_ELL_THETA_CAP;
            if (const char *d = getenv("KH_ELL_CAP")) cap = atof(d) > 0.0 && atof(d) < cap ? atof(d) : cap;  // (as theta_max above)
            kh_build_real_spectrum_rows(e->tol, tab.data(), c0.data(), rows.data(), ratios.data(), cap,
                                        e->real_spectrum ? 0.0 : e->imag_defect);
        } else if (e->real_spectrum) {
            kh_build_real_spectrum_rows(e->tol, tab.data(), c0.data(), rows.data(), ratios.data());
        } else if (e->imag_defect > 0.0 && e->imag_defect <= 0.05 && !(getenv("KH_NEAR_IMAG") && atoi(getenv("KH_NEAR_IMAG")) == 0)) {
            // the same form, with the margin for the Hermitian defect, for the other kernel families (weakly damped
            // Liouvillians, Hamiltonians with a small anti-Hermitian part); KH_NEAR_IMAG=0: Taylor (A/B switch)
            kh_build_real_spectrum_rows(e->tol, tab.data(), c0.data(), rows.data(), ratios.data(), 2.0, e->imag_defect);
        } else {
            kh_build_degree_table(e->tol, tab.data());
            kh_build_taylor_rows(c0.data(), rows.data(), ratios.data());
        }
        KH_HIP_E(hipMalloc(&e->d_ratios, sizeof(double) * ratios.size()));
        KH_HIP_E(hipMemcpy(e->d_ratios, ratios.data(), sizeof(double) * ratios.size(), hipMemcpyHostToDevice));
        KH_HIP_E(hipMalloc(&e->d_q2_theta, sizeof(double) * tab.size()));
        KH_HIP_E(hipMalloc(&e->d_q2_c0, sizeof(double) * c0.size()));
        KH_HIP_E(hipMalloc(&e->d_q2_rows, sizeof(double) * rows.size()));
        KH_HIP_E(hipMemcpy(e->d_q2_theta, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
        KH_HIP_E(hipMemcpy(e->d_q2_c0, c0.data(), sizeof(double) * c0.size(), hipMemcpyHostToDevice));
        KH_HIP_E(hipMemcpy(e->d_q2_rows, rows.data(), sizeof(double) * rows.size(), hipMemcpyHostToDevice));
    }
    if (e->kind_store == KIND_TILE_Q2)
        KH_HIP_E(hipFuncSetAttribute((const void *)kh_q2_sweep_store, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)kh_q2_lds_bytes()));
    if (e->kind == KIND_TILE_Q2) {
        KH_HIP_E(hipFuncSetAttribute((const void *)kh_q2_forward_update<false, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kh_q2_lds_bytes()));
        KH_HIP_E(hipFuncSetAttribute((const void *)kh_q2_forward_update<false, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kh_q2_lds_bytes()));
        KH_HIP_E(hipFuncSetAttribute((const void *)kh_q2_forward_update<false, true, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kh_q2_lds_bytes()));
        KH_HIP_E(hipFuncSetAttribute((const void *)kh_q2_forward_update<false, false, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kh_q2_lds_bytes()));
        KH_HIP_E(hipFuncSetAttribute((const void *)kh_q2_forward_update<true, false, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kh_q2_lds_bytes()));
        KH_HIP_E(hipFuncSetAttribute((const void *)kh_q2_forward_update<true, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kh_q2_lds_bytes()));

    }

    if (e->kind == KIND_TILE_Q2 && !e->mini && e->K > 1) {
        // every workgroup of the single-launch update sweep must be resident at once: ask the occupancy of the kernel
        // as built (registers, LDS) instead of assuming one per CU; if it does not fit, the generic kernels (which
        // loop over objectives inside at most #CUs workgroups) take over
        // (every instantiation a sweep of this engine may launch: one GPU / sharded, first / second order, sums on
        // either side -- their register footprints differ)
        const void *forms[] = {(const void *)kh_q2_forward_update<false, true, true>, (const void *)kh_q2_forward_update<false, false, true>,
                               (const void *)kh_q2_forward_update<true, false, true>, (const void *)kh_q2_forward_update<false, true>,
                               (const void *)kh_q2_forward_update<false, false>,      (const void *)kh_q2_forward_update<true, false>};
        int rc = KH_OK;
        for (const void *f : forms)
            if (rc == KH_OK) rc = check_residency(e, f, KH_Q2_THREADS, kh_q2_lds_bytes(), e->K, "kh_q2_forward_update");
        if (rc != KH_OK) {
            e->kind = KIND_GENERIC;  // (the plain sweeps keep the q2 kernel: its workgroups do not wait for each other)
            e->grid_update = e->K < max_wgs ? e->K : max_wgs;
        }
    }

    // ---- ensembles (kh_ens.h): one drift, control operators equal up to a real scale, N <= 64, one control, more
    // objectives than CUs (from 257 on it beats the two-workgroups-per-CU tile kernels too: 10.7 against 11.1 us per interval at
    // K = 512).  KH_ENS=0: off;
    // KH_ENS=1: for any K (testing); KH_ENS_MINK: smallest K that takes it; KH_ENS_NCG: column groups (testing)
    {
        const char *ens_env = getenv("KH_ENS");
        const int ens_mode = ens_env ? atoi(ens_env) : -1;
        int min_k = 257;  // (one objective per CU: the two-terms-per-phase kernels, 4.8 us per interval against 10.2 here)
        if (const char *d = getenv("KH_ENS_MINK")) min_k = atoi(d);
        const bool want = ens_mode == 1 || (ens_mode != 0 && force == nullptr && e->K >= min_k);
        if (want && csr_fw == nullptr && e->N <= KH_TILE_N && e->L == 1 && fw[1] != nullptr) {
            int ncg = 0;
            for (int c = 1; c <= KH_ENS_MAXCG; c *= 2)
                if ((e->K + 2 * c - 1) / (2 * c) <= max_wgs) {
                    ncg = c;
                    break;
                }
            if (const char *d = getenv("KH_ENS_NCG")) {
                const int c = atoi(d);
                if ((c == 1 || c == 2 || c == 4 || c == 8) && (e->K + 2 * c - 1) / (2 * c) <= max_wgs) ncg = c;
            }
            if (ncg > 0) {
                // reference element: the largest component of objective 0's control operator
                std::vector<cplx> ref((size_t)e->N * e->N);
                KH_HIP_E(hipMemcpy(ref.data(), fw[1], sizeof(cplx) * ref.size(), hipMemcpyDeviceToHost));
                int ref_idx = 0, ref_comp = 0;
                double best = 0.0;
                for (size_t i = 0; i < ref.size(); ++i) {
                    if (fabs(ref[i].x) > best) best = fabs(ref[i].x), ref_idx = (int)i, ref_comp = 0;
                    if (fabs(ref[i].y) > best) best = fabs(ref[i].y), ref_idx = (int)i, ref_comp = 1;
                }
                if (best > 0.0) {
                    int *d_flags = nullptr, flags[2] = {1, 1};
                    KH_HIP_E(hipMalloc(&e->d_ens_scale, sizeof(double) * e->K));
                    KH_HIP_E(hipMalloc(&d_flags, sizeof(flags)));
                    hipError_t err = hipMemset(d_flags, 0, sizeof(flags));
                    if (err == hipSuccess) {
                        kh_ens_detect_kernel<<<e->K, 256>>>(e->d_ops_fw, e->K, e->N, ref_idx, ref_comp, e->d_ens_scale, d_flags);
                        err = hipGetLastError();  // (the launch's own verdict, not whatever the copy below reports)
                        if (err == hipSuccess) err = hipMemcpy(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost);
                    }
                    (void)hipFree(d_flags);
                    KH_HIP_E(err);
                    if (flags[0] == 0 && flags[1] == 0) {
                        e->ens = true;
                        e->ens2 = !(getenv("KH_ENS2") && atoi(getenv("KH_ENS2")) == 0);
                        e->ens_ncg = ncg;
                        e->ens_G = (e->K + 2 * ncg - 1) / (2 * ncg);
                        e->ens_H0 = fw[0];
                        e->ens_H1 = fw[1];
                    }
                }
            }
        }
        if (e->ens) {  // all ens_G workgroups resident at once?  (as the q2 path asks for its instantiations)
            const void *forms[2] = {nullptr, nullptr};
            switch (e->ens_ncg) {
                case 1: forms[0] = (const void *)kh_ens_forward_update<1, false>, forms[1] = (const void *)kh_ens_forward_update<1, true>; break;
                case 2: forms[0] = (const void *)kh_ens_forward_update<2, false>, forms[1] = (const void *)kh_ens_forward_update<2, true>; break;
                case 4: forms[0] = (const void *)kh_ens_forward_update<4, false>, forms[1] = (const void *)kh_ens_forward_update<4, true>; break;
                default: forms[0] = (const void *)kh_ens_forward_update<8, false>, forms[1] = (const void *)kh_ens_forward_update<8, true>; break;
            }
            int rc = KH_OK;
            for (const void *f : forms) {
                if (rc == KH_OK) rc = ensure_dynamic_lds(e, f, kh_ens_lds_bytes(e->ens_ncg));
                if (rc == KH_OK) rc = check_residency(e, f, KH_ENS_THREADS, kh_ens_lds_bytes(e->ens_ncg), e->ens_G, "kh_ens_forward_update");
            }
            if (rc != KH_OK) e->ens = false;
        }
    }

    // ---- workspaces
    KH_HIP_E(hipMalloc(&e->d_phi, sizeof(cplx) * (size_t)e->K * e->N));
    const int Lx = e->L > 0 ? e->L : 1;
    const int slot_wgs = e->kind == KIND_COOP && e->coop_G * e->coop_Y > e->grid_update ? e->coop_G * e->coop_Y
                                                                                         : e->grid_update;
    e->slots_bytes = sizeof(kh_u64) * 2 * (size_t)slot_wgs * Lx * 2;
    KH_HIP_E(hipMalloc(&e->d_slots, e->slots_bytes));
    KH_HIP_E(hipMalloc(&e->d_abort, 2 * sizeof(unsigned int)));  // [0] abort flag, [1] (KH_TIMING) polling rounds
    KH_HIP_E(hipMemset(e->d_abort, 0, 2 * sizeof(unsigned int)));
    KH_HIP_E(hipMalloc(&e->d_wait_ticks, 4 * sizeof(unsigned long long)));
    KH_HIP_E(hipMemset(e->d_wait_ticks, 0, 4 * sizeof(unsigned long long)));
    KH_HIP_E(hipMalloc(&e->d_stats, sizeof(double) * 68));
    KH_HIP_E(hipMemset(e->d_stats, 0, sizeof(double) * 68));
    KH_HIP_E(hipMalloc(&e->d_wg_partial, sizeof(double) * (size_t)e->grid_update * Lx));
    KH_HIP_E(hipMalloc(&e->d_step_partial, sizeof(double) * Lx));
    KH_HIP_E(hipDeviceSynchronize());
#undef KH_HIP_E
    *out = e;
    return KH_OK;
}

extern "C" int kh_engine_create(const kh_problem *pr, kh_engine **out) {
    return engine_create(pr, nullptr, nullptr, out);
}

extern "C" int kh_engine_create_csr(const kh_problem_csr *pc, kh_engine **out) {
    if (pc == nullptr || out == nullptr) return kh_fail(KH_ERR_INVALID, "null argument");
    *out = nullptr;
    if (pc->ops == nullptr || pc->ops_adj == nullptr || pc->op_norms == nullptr)
        return kh_fail(KH_ERR_INVALID, "ops, ops_adj and op_norms are required for sparse operators");
    if (pc->K < 1 || pc->L < 0) return kh_fail(KH_ERR_INVALID, "bad sizes K=%d L=%d", pc->K, pc->L);
    const size_t nops = (size_t)pc->K * (1 + pc->L);
    std::vector<const kh_cdouble *> data(nops);
    for (size_t i = 0; i < nops; ++i) {
        const kh_csr &a = pc->ops[i], &b = pc->ops_adj[i];
        if ((a.data == nullptr) != (b.data == nullptr))
            return kh_fail(KH_ERR_INVALID, "operator %zu: ops and ops_adj must both be present or both absent", i);
        if (a.data != nullptr && (a.indptr == nullptr || a.indices == nullptr || b.indptr == nullptr ||
                                  b.indices == nullptr || a.nnz != b.nnz))
            return kh_fail(KH_ERR_INVALID, "operator %zu: incomplete CSR arrays", i);
        data[i] = a.data;
    }
    kh_problem pr;
    pr.K = pc->K;
    pr.N = pc->N;
    pr.L = pc->L;
    pr.nt = pc->nt;
    pr.is_super = pc->is_super;
    pr.reserved = 0;
    pr.dt = pc->dt;
    pr.ops = data.data();
    pr.op_norms = pc->op_norms;
    pr.tol = pc->tol;
    pr.theta_max = pc->theta_max;
    return engine_create(&pr, pc->ops, pc->ops_adj, out);
}

// ---------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------

template <int RPT, int LT>
static int launch_tile_store(kh_engine *e, const KhSweepArgs &p, const double *pulses, const cplx *in,
                             cplx *store, cplx *out, int direction, hipStream_t st) {
    constexpr size_t lds = KhTileLds<RPT, LT>::bytes(KhTileLds<RPT, LT>::STORE);  // operator tiles parked in LDS
    const int rc = ensure_dynamic_lds(e, (const void *)kh_tile_sweep_store<RPT, LT>, lds);
    if (rc != KH_OK) return rc;
    launch_plain<kh_tile_sweep_store<RPT, LT>>(dim3(e->K), dim3(512 / RPT), lds, st, p, pulses, in, store, out, direction);
    return KH_OK;
}

template <int RPT>
static int dispatch_tile_store(kh_engine *e, const KhSweepArgs &p, const double *pulses, const cplx *in,
                               cplx *store, cplx *out, int direction, hipStream_t st) {
    switch (e->L) {
        case 1: return launch_tile_store<RPT, 1>(e, p, pulses, in, store, out, direction, st);
        case 2: return launch_tile_store<1, 2>(e, p, pulses, in, store, out, direction, st);
        case 3: return launch_tile_store<1, 3>(e, p, pulses, in, store, out, direction, st);
        case 4: return launch_tile_store<1, 4>(e, p, pulses, in, store, out, direction, st);
        default: return kh_fail(KH_ERR_UNSUPPORTED, "tile kernels handle 1..4 controls");
    }
}

static KhExchange exchange_args(const kh_engine *e, bool internal_exchange);

static KhCoopArgs coop_args(const kh_engine *e, bool backward) {
    KhCoopArgs c;
    c.fops = backward ? e->d_coop_fops_bw : e->d_coop_fops_fw;
    c.sq = backward ? e->d_coop_sq_bw : e->d_coop_sq_fw;  // (NULL unless staged: one control)
    c.vbuf = e->d_coop_vbuf;
    c.epoch_base = 0;  // the buffer is cleared before every launch
    c.G = e->coop_G;
    c.Y = e->coop_Y;
    c.ks = e->coop_ks;
    c.cols = e->coop_cols;
    c.first_poll_delay = e->coop_poll_delay;
    c.xcd_rows = e->coop_xcd ? e->coop_G : 0;
    c.xcc = e->d_coop_xcc;
    c.local = 0;
    c.ring_mask = KH_COOP_RING - 1;
    for (int i = 0; i < 5; ++i) c.tab[i] = nullptr;  // (resolved in the kernel)
    c.ser_theta = e->coop_series ? e->d_q2_theta : nullptr;
    c.ser_c0 = e->coop_series ? e->d_q2_c0 : nullptr;
    c.ser_rows = e->coop_series ? e->d_q2_rows : nullptr;
    return c;
}

// The cooperative kernels are launched with one column group per XCD where that is possible: a one-dimensional
// grid of 8 G blocks of which only G Y do anything (kh_coop_place).  The cooperative-launch validation counts all
// 8 G of them, so on a device (or partition, or CU mask) with fewer resident workgroups than that the launch is
// refused although the G Y real ones would fit: the placement is then given up for good (two-dimensional (G, Y)
// grid, memory-side exchange) and the launch repeated.
template <class Launch>
static int launch_coop_placed(kh_engine *e, Launch &&launch) {
    int rc = launch(e->coop_xcd ? dim3(8 * e->coop_G) : dim3(e->coop_G, e->coop_Y));
    if (rc == KH_ERR_UNSUPPORTED && e->coop_xcd) {
        e->coop_xcd = false;
        rc = launch(dim3(e->coop_G, e->coop_Y));
    }
    return rc;
}

#ifdef KH_WITH_C4W
static KhCoopArgs c4_args(const kh_engine *e, bool backward) {
    KhCoopArgs c = coop_args(e, backward);
    c.fops = backward ? e->d_c4_fops_bw : e->d_c4_fops_fw;
    c.sq = backward ? e->d_c4_sq_bw : e->d_c4_sq_fw;
    c.ks = e->c4_NG;
    return c;
}

template <int MAXG>
static int launch_c4_store(kh_engine *e, const KhSweepArgs &p, const double *pulses, const cplx *in, cplx *store,
                           cplx *out, int direction, hipStream_t st) {
    const size_t lds = kh_c4_lds_bytes(MAXG);
    const int rc = ensure_dynamic_lds(e, (const void *)kh_c4_sweep_store<MAXG>, lds);
    if (rc != KH_OK) return rc;
    KH_HIP(hipMemsetAsync(e->d_coop_vbuf, 0, e->coop_vbuf_bytes, st));
    KH_HIP(hipMemsetAsync(e->d_coop_xcc, 0, sizeof(unsigned int) * (size_t)e->coop_G * e->coop_Y, st));
    return launch_coop_placed(e, [&](dim3 grid) {
        return launch_persistent<kh_c4_sweep_store<MAXG>>(e, grid, dim3(KH_C4_THREADS), lds, st, p, c4_args(e, direction < 0),
                                 exchange_args(e, true), pulses, in, store, out, direction);
    });
}

#endif

template <int MAXKS, int COLS>
static int launch_coop_store(kh_engine *e, const KhSweepArgs &p, const double *pulses, const cplx *in, cplx *store,
                             cplx *out, int direction, hipStream_t st) {
    // (16 operator slots per lane x 16 objectives per workgroup: the A^2 chain's second resident fragment does not fit
    // the register file -- 245 .. 343 spilled values --, the engine does not stage the A^2 tables for that shape)
    constexpr bool SQ_FORMS = !(MAXKS == 16 && COLS == 16);
    const bool sq = SQ_FORMS && (direction < 0 ? e->d_coop_sq_bw : e->d_coop_sq_fw) != nullptr;
    int rc = KH_OK;
    if constexpr (SQ_FORMS)
        rc = ensure_dynamic_lds(e, (const void *)kh_coop_sweep_store<MAXKS, COLS, true>, kh_coop_lds_bytes(COLS <= 4 ? 16 : 15, COLS));
    if (rc == KH_OK)
        rc = ensure_dynamic_lds(e, (const void *)kh_coop_sweep_store<MAXKS, COLS, false>, kh_coop_lds_bytes(COLS <= 4 ? 16 : 15, COLS));
    if (rc != KH_OK) return rc;
    KH_HIP(hipMemsetAsync(e->d_coop_vbuf, 0, e->coop_vbuf_bytes, st));
    KH_HIP(hipMemsetAsync(e->d_coop_xcc, 0, sizeof(unsigned int) * (size_t)e->coop_G * e->coop_Y, st));
    return launch_coop_placed(e, [&](dim3 grid) {
        if constexpr (SQ_FORMS)
            if (sq)
                return launch_persistent<kh_coop_sweep_store<MAXKS, COLS, true>>(e, grid, dim3(KH_COOP_THREADS), kh_coop_lds_bytes(e->coop_ks, COLS), st, p,
                                         coop_args(e, direction < 0), exchange_args(e, true), pulses, in, store, out, direction);
        return launch_persistent<kh_coop_sweep_store<MAXKS, COLS, false>>(e, grid, dim3(KH_COOP_THREADS), kh_coop_lds_bytes(e->coop_ks, COLS), st, p,
                                 coop_args(e, direction < 0), exchange_args(e, true), pulses, in, store, out, direction);
    });
}

template <int MAXKS, int COLS>
static int launch_coop_update(kh_engine *e, const KhSweepArgs &p, const KhUpdateArgs &u_in, const KhExchange &ex,
                              hipStream_t st) {
    KhUpdateArgs u = u_in;
    if (e->coop_adj && u.sigma == nullptr && e->L == 1 && e->d_coop_sq_fw != nullptr && e->d_coop_adj == nullptr) {
        // V = H_1^+ X needs a second buffer of the co-state store's size; it is an optimisation (one round per interval
        // less): without the memory the sweep takes the sums by one more round, as it did before the form existed
        const size_t bytes = sizeof(cplx) * (size_t)e->K * e->nt * e->N;
        if (hipMalloc(&e->d_coop_adj, bytes) != hipSuccess) {
            (void)hipGetLastError();
            e->d_coop_adj = nullptr;
            e->coop_adj = false;
        }
    }
    constexpr bool SQ_FORMS = !(MAXKS == 16 && COLS == 16);  // (see launch_coop_store)
    const bool sq = SQ_FORMS && e->d_coop_sq_fw != nullptr;
    const bool adj = e->coop_adj && u.sigma == nullptr && e->L == 1 && sq;
    const void *func = u.sigma != nullptr ? (const void *)kh_coop_forward_update<MAXKS, COLS, true, false, false>
                                          : (const void *)kh_coop_forward_update<MAXKS, COLS, false, false, false>;
    if constexpr (SQ_FORMS) {
        if (u.sigma != nullptr && sq)
            func = (const void *)kh_coop_forward_update<MAXKS, COLS, true, false, true>;
        else if (adj)
            func = ex.world == 1 && e->coop_single ? (const void *)kh_coop_forward_update<MAXKS, COLS, false, true, true, false>
                                                   : (const void *)kh_coop_forward_update<MAXKS, COLS, false, true, true>;
        else if (u.sigma == nullptr && sq)
            func = (const void *)kh_coop_forward_update<MAXKS, COLS, false, false, true>;
    }
    const int rc = ensure_dynamic_lds(e, func, kh_coop_lds_bytes(COLS <= 4 ? 16 : 15, COLS));
    if (rc != KH_OK) return rc;
    if (adj) {
        // V = H_1^+ X over the whole co-state store, block-sparse on the matrix cores (kh_coop.h)
        const long long M = (long long)e->K * e->nt;
        const unsigned blocks = (unsigned)((M + 63) / 64);
        kh_coop_adjoint_side<<<blocks, KH_COOP_ADJ_THREADS, 0, st>>>(e->coop_adj_op, e->d_coop_adj_nz, u.chi_store, e->d_coop_adj,
                                                                     e->N, e->coop_G, M);
        KH_HIP(hipGetLastError());
        u.adj_store = e->d_coop_adj;
    }
    KH_HIP(hipMemsetAsync(e->d_coop_vbuf, 0, e->coop_vbuf_bytes, st));
    KH_HIP(hipMemsetAsync(e->d_coop_xcc, 0, sizeof(unsigned int) * (size_t)e->coop_G * e->coop_Y, st));
    return launch_coop_placed(e, [&](dim3 grid) {
        const size_t lds = kh_coop_lds_bytes(e->coop_ks, COLS);
        const KhCoopArgs ca = coop_args(e, false);
        if constexpr (SQ_FORMS) {
            if (adj && ex.world == 1 && e->coop_single)
                return launch_persistent<kh_coop_forward_update<MAXKS, COLS, false, true, true, false>>(e, grid, dim3(KH_COOP_THREADS), lds, st, p, ca, u, ex);
            if (adj) return launch_persistent<kh_coop_forward_update<MAXKS, COLS, false, true, true>>(e, grid, dim3(KH_COOP_THREADS), lds, st, p, ca, u, ex);
            if (u.sigma != nullptr && sq)
                return launch_persistent<kh_coop_forward_update<MAXKS, COLS, true, false, true>>(e, grid, dim3(KH_COOP_THREADS), lds, st, p, ca, u, ex);
            if (u.sigma == nullptr && sq)
                return launch_persistent<kh_coop_forward_update<MAXKS, COLS, false, false, true>>(e, grid, dim3(KH_COOP_THREADS), lds, st, p, ca, u, ex);
        }
        if (u.sigma != nullptr)
            return launch_persistent<kh_coop_forward_update<MAXKS, COLS, true, false, false>>(e, grid, dim3(KH_COOP_THREADS), lds, st, p, ca, u, ex);
        return launch_persistent<kh_coop_forward_update<MAXKS, COLS, false, false, false>>(e, grid, dim3(KH_COOP_THREADS), lds, st, p, ca, u, ex);
    });
}

static int sweep_store(kh_engine *e, bool backward, const double *pulses, const cplx *in, cplx *store, cplx *out,
                       hipStream_t st) {
    const KhSweepArgs p = sweep_args(e, backward);
    const int direction = backward ? -1 : +1;
    KH_HIP(hipMemsetAsync(e->d_stats, 0, sizeof(double) * 4, st));
    int rc = KH_OK;
    if (e->kind_store == KIND_TILE_Q2 && e->quad) {
        launch_plain<kh_quad_sweep_store>(dim3(1), dim3(64), 0, st, p, backward ? e->d_sq_bw : e->d_sq_fw, pulses, in, store, out, direction);
    } else if (e->kind_store == KIND_TILE_Q2 && e->mini) {
        launch_plain<kh_mini_sweep_store>(dim3(e->K), dim3(64), 0, st, p, backward ? e->d_sq_bw : e->d_sq_fw, pulses, in, store, out, direction);
#ifdef KH_WITH_Q4
    } else if (e->kind_store == KIND_TILE_Q2 && e->use_q4) {
        rc = ensure_dynamic_lds(e, (const void *)kh_q4_sweep_store, kh_q4_lds_bytes());
        if (rc == KH_OK)
            launch_plain<kh_q4_sweep_store>(dim3(e->K), dim3(KH_Q4_THREADS), kh_q4_lds_bytes(), st, 
                p, backward ? e->d_sq_bw : e->d_sq_fw, pulses, in, store, out, direction);
#endif
    } else if (e->kind_store == KIND_TILEN) {
        const cplx *const *tabs = backward ? e->d_tn_bw : e->d_tn_fw;
        const int grid = e->K < e->num_cus ? e->K : e->num_cus;
        const size_t lds = kh_tn_lds_bytes();
#define KH_TN_STORE(EP, HR) launch_plain<kh_tn_sweep_store<EP, HR>>(dim3(grid), dim3(KH_TN_THREADS), lds, st, p, tabs, pulses, in, store, out, direction)
        if (e->N <= 80) {
            if (e->tn_h1reg) KH_TN_STORE(20, true); else KH_TN_STORE(20, false);
        } else if (e->N <= 96) {
            if (e->tn_h1reg) KH_TN_STORE(24, true); else KH_TN_STORE(24, false);
        } else if (e->N <= 112) {
            KH_TN_STORE(28, false);
        } else {
            KH_TN_STORE(32, false);
        }
#undef KH_TN_STORE
    } else if (e->kind_store == KIND_TILEX) {
        const int grid = e->K < e->num_cus ? e->K : e->num_cus;
        const cplx *const *tabs = backward ? e->d_tx_bw : e->d_tx_fw;
#define KH_TX_STORE(LT)                                                                                                    \
    do {                                                                                                                   \
        rc = ensure_dynamic_lds(e, (const void *)kh_tx_sweep_store<LT>, kh_tx_lds_bytes());                                \
        if (rc == KH_OK)                                                                                                   \
            launch_plain<kh_tx_sweep_store<LT>>(dim3(grid), dim3(KH_TX_THREADS), kh_tx_lds_bytes(), st, p, tabs, pulses, in, \
                                                store, out, direction);                                                    \
    } while (0)
        switch (e->L) {
            case 5: KH_TX_STORE(5); break;
            case 6: KH_TX_STORE(6); break;
            case 7: KH_TX_STORE(7); break;
            default: KH_TX_STORE(8); break;
        }
#undef KH_TX_STORE
    } else if (e->kind_store == KIND_ELL) {
        const KhEll *ells = backward ? e->d_ell_bw : e->d_ell_fw;
        const int grid = e->ell_stream ? (e->K < e->num_cus ? e->K : e->num_cus) : (e->K < 4 * e->num_cus ? e->K : 4 * e->num_cus);
        const size_t lds = kh_ell_lds_bytes(e->ell_stream);
        // (two vector buffers of KH_ELL_NMAX elements: more than the 64 KiB a kernel gets without asking)
#define KH_ELL_STORE(T, R, EM)                                                                                        \
    do {                                                                                                              \
        rc = ensure_dynamic_lds(e, (const void *)kh_ell_sweep_store<T, R, EM>, lds);                                  \
        if (rc == KH_OK)                                                                                              \
            launch_plain<kh_ell_sweep_store<T, R, EM>>(dim3(grid), dim3(T), lds, st, p, ells, e->d_ell_off, e->d_ell_vals, \
                                                       pulses, in, store, out, direction, (cplx *)nullptr, 0LL);     \
    } while (0)
        // one row per lane where the rows' entries fit the register budget of that many waves (512 threads: 256 VGPRs,
        // 768: 168, 1024: 128), else two rows per lane of a 512-thread workgroup
        if (e->ell_stream) {
            rc = ensure_dynamic_lds(e, (const void *)kh_ell_sweep_store<512, KH_ELLS_RPL, 4, true>, lds);
            if (rc == KH_OK)
                launch_plain<kh_ell_sweep_store<512, KH_ELLS_RPL, 4, true>>(dim3(grid), dim3(512), lds, st, p, ells, e->d_ell_off,
                                                                             e->d_ell_vals, pulses, in, store, out, direction,
                                                                             e->d_ell_scratch, e->ell_scratch_stride);
        } else if (e->N <= 512) {
            if (e->ell_E <= 8) KH_ELL_STORE(512, 1, 8);
            else if (e->ell_E <= 12) KH_ELL_STORE(512, 1, 12);
            else if (e->ell_E <= 16) KH_ELL_STORE(512, 1, 16);
            else if (e->ell_E <= 24) KH_ELL_STORE(512, 1, 24);
            else KH_ELL_STORE(512, 1, 32);
        } else if (e->N <= 768 && e->ell_E <= 16) {
            // (12: drift + two controls of a Lindbladian -- the reference's notebook 06 has 11.2 entries per row; every
            // padded slot is a gather and four multiply-adds per term)
            if (e->ell_E <= 8) KH_ELL_STORE(768, 1, 8);
            else if (e->ell_E <= 12) KH_ELL_STORE(768, 1, 12);
            else KH_ELL_STORE(768, 1, 16);
        } else if (e->N > 1024) {  // (<= 8 entries per row: build_ell_host)
            if (e->N <= 1536) KH_ELL_STORE(512, 3, 8);
            else KH_ELL_STORE(512, 4, 8);
        } else if (e->ell_E <= 8) {
            KH_ELL_STORE(1024, 1, 8);
        } else {
            if (e->ell_E <= 12) KH_ELL_STORE(512, 2, 12);
            else KH_ELL_STORE(512, 2, 16);
        }
#undef KH_ELL_STORE
    } else if (e->kind_store == KIND_TILE_Q2) {
        launch_plain<kh_q2_sweep_store>(dim3(e->K), dim3(KH_Q2_THREADS), kh_q2_lds_bytes(), st, 
            p, backward ? e->d_sq_bw : e->d_sq_fw, pulses, in, store, out, direction);
    } else if (e->kind_store == KIND_TILE_RPT2) {
        rc = dispatch_tile_store<2>(e, p, pulses, in, store, out, direction, st);
    } else if (e->kind_store == KIND_TILE_RPT1) {
        rc = dispatch_tile_store<1>(e, p, pulses, in, store, out, direction, st);
#ifdef KH_WITH_C4W
    } else if (e->kind_store == KIND_COOP && e->coop4w) {
        rc = e->c4_NG <= 8    ? launch_c4_store<8>(e, p, pulses, in, store, out, direction, st)
             : e->c4_NG <= 16 ? launch_c4_store<16>(e, p, pulses, in, store, out, direction, st)
             : e->c4_NG <= 26 ? launch_c4_store<26>(e, p, pulses, in, store, out, direction, st)
                              : launch_c4_store<30>(e, p, pulses, in, store, out, direction, st);
        if (rc == KH_ERR_UNSUPPORTED) {
            e->kind_store = KIND_GENERIC;
            return sweep_store(e, backward, pulses, in, store, out, st);
        }
#endif
    } else if (e->kind_store == KIND_COOP) {
        if (e->coop_cols == 2)
            rc = e->coop_ks <= 8 ? launch_coop_store<8, 2>(e, p, pulses, in, store, out, direction, st)
                                 : launch_coop_store<16, 2>(e, p, pulses, in, store, out, direction, st);
        else if (e->coop_cols == 4)
            rc = e->coop_ks <= 8 ? launch_coop_store<8, 4>(e, p, pulses, in, store, out, direction, st)
                                 : launch_coop_store<16, 4>(e, p, pulses, in, store, out, direction, st);
        else
            rc = e->coop_ks <= 8 ? launch_coop_store<8, 16>(e, p, pulses, in, store, out, direction, st)
                                 : launch_coop_store<16, 16>(e, p, pulses, in, store, out, direction, st);
        if (rc == KH_ERR_UNSUPPORTED) {
            // not even the G x Y grid can be resident at once (fewer CUs than expected): the plain sweeps need no
            // cross-workgroup exchange at all, so the per-objective generic kernel takes them over from here on
            e->kind_store = KIND_GENERIC;
            return sweep_store(e, backward, pulses, in, store, out, st);
        }
    } else {
        if (!e->gen_fits) return kh_fail(KH_ERR_UNSUPPORTED, "N=%d: the generic kernels' vectors do not fit LDS", e->N);
        const size_t lds = kh_gen_lds_bytes(e->N, e->d_csr_fw == nullptr);
        rc = ensure_dynamic_lds(e, (const void *)kh_gen_sweep_store, lds);
        // (the workgroups loop over the objectives: no more of them than the device runs at once -- each may own an
        // N x N scratch generator)
        const int grid = e->K < 2 * e->num_cus ? e->K : 2 * e->num_cus;
        ensure_gen_scratch(e, grid);
        KhSweepArgs pg = p;
        pg.gen_scratch = e->d_gen_scratch;
        pg.gen_scratch_wgs = e->gen_scratch_wgs;
        if (rc == KH_OK) launch_plain<kh_gen_sweep_store>(dim3(grid), dim3(KH_GEN_THREADS), lds, st, pg, pulses, in, store, out, direction);
    }
    if (rc != KH_OK) return rc;
    KH_HIP(hipGetLastError());
    e->last_intervals = e->nt - 1;
    e->last_wgs = e->K;
    return KH_OK;
}

extern "C" int kh_forward_store(kh_engine *e, const double *pulses_dev, const kh_cdouble *init_dev,
                                kh_cdouble *states_dev, kh_cdouble *psi_T_dev, void *stream) {
    if (e == nullptr || pulses_dev == nullptr && e->L > 0 || init_dev == nullptr)
        return kh_fail(KH_ERR_INVALID, "null argument");
    return sweep_store(e, false, pulses_dev, (const cplx *)init_dev, (cplx *)states_dev, (cplx *)psi_T_dev,
                       (hipStream_t)stream);
}

extern "C" int kh_backward_store(kh_engine *e, const kh_cdouble *chi_T_dev, const double *pulses_dev,
                                 kh_cdouble *chi_store_dev, void *stream) {
    if (e == nullptr || chi_T_dev == nullptr || chi_store_dev == nullptr)
        return kh_fail(KH_ERR_INVALID, "null argument");
    return sweep_store(e, true, pulses_dev, (const cplx *)chi_T_dev, (cplx *)chi_store_dev, nullptr,
                       (hipStream_t)stream);
}

template <int RPT, int LT>
static int launch_tile_update(kh_engine *e, const KhSweepArgs &p, const KhUpdateArgs &u, const KhExchange &ex,
                              hipStream_t st) {
    constexpr size_t lds = KhTileLds<RPT, LT>::bytes(KhTileLds<RPT, LT>::UPDATE);
    const void *func = u.sigma != nullptr ? (const void *)kh_tile_forward_update<RPT, LT, true>
                                          : (const void *)kh_tile_forward_update<RPT, LT, false>;
    const int rc = ensure_dynamic_lds(e, func, lds);
    if (rc != KH_OK) return rc;
    if (!u.internal_exchange) {  // one launch per interval (sharded sweep): nothing waits inside the kernel
        if (u.sigma != nullptr)
            launch_plain<kh_tile_forward_update<RPT, LT, true>>(dim3(e->K), dim3(512 / RPT), lds, st, p, u, ex);
        else
            launch_plain<kh_tile_forward_update<RPT, LT, false>>(dim3(e->K), dim3(512 / RPT), lds, st, p, u, ex);
        return KH_OK;
    }
    KhExchange exl = ex;
    // several controls: L waves gather side by side (kh_tile64.h) and a failed polling round costs L times the loads,
    // so the stores get a longer head start -- measured best on config-5 shapes: 24 / 28 / 32 x 64 cycles for L = 2 / 3 / 4
    // (update sweep 7.47 / - / 9.86 us per interval against 7.92 / - / 11.21 with 16)
    if (LT >= 2 && !e->poll_delay_set) exl.first_poll_delay = 24 + 4 * (LT - 2);
    if (exl.world == 1 && exl.G > 1 && e->tile_single) {
        const void *fs = u.sigma != nullptr ? (const void *)kh_tile_forward_update<RPT, LT, true, true>
                                            : (const void *)kh_tile_forward_update<RPT, LT, false, true>;
        const int rcs = ensure_dynamic_lds(e, fs, lds);
        if (rcs != KH_OK) return rcs;
        if (u.sigma != nullptr)
            return launch_persistent<kh_tile_forward_update<RPT, LT, true, true>>(e, dim3(e->K), dim3(512 / RPT), lds, st, p, u, exl);
        return launch_persistent<kh_tile_forward_update<RPT, LT, false, true>>(e, dim3(e->K), dim3(512 / RPT), lds, st, p, u, exl);
    }
    if (u.sigma != nullptr)
        return launch_persistent<kh_tile_forward_update<RPT, LT, true>>(e, dim3(e->K), dim3(512 / RPT), lds, st, p, u, exl);
    return launch_persistent<kh_tile_forward_update<RPT, LT, false>>(e, dim3(e->K), dim3(512 / RPT), lds, st, p, u, exl);
}

static KhExchange exchange_args(const kh_engine *e, bool internal_exchange) {
    KhExchange ex;
    ex.slots = e->d_slots;
    ex.abort_flag = e->d_abort;
    ex.G = (e->kind == KIND_COOP && internal_exchange) ? e->coop_G * e->coop_Y : e->grid_update;
    ex.timeout_ticks = e->timeout_ticks;
    // across GPUs the ranks are separate processes: a host-side hiccup of one of them (garbage collection, page
    // faults) must not look like a lost peer and demote the whole run to the per-interval path
    if (e->p2p_ready && internal_exchange && !e->timeout_set && ex.timeout_ticks < 1000000000LL)
        ex.timeout_ticks = 1000000000LL;  // 10 s
    ex.peer_windows = e->d_p2p_peers;
    ex.my_window = e->p2p_window;
    ex.world = (e->p2p_ready && internal_exchange) ? e->p2p_world : 1;
    ex.rank = e->p2p_rank;
    ex.epoch_base = e->p2p_epoch_base;
    ex.first_poll_delay = e->poll_delay;
    ex.fail_at = (ex.world > 1 && e->p2p_rank == e->p2p_fail_rank && e->p2p_sweeps + 1 == e->p2p_fail_sweep) ? e->p2p_fail_at : -1;
    ex.wait_ticks = ex.world > 1 ? e->d_wait_ticks : nullptr;
    return ex;
}

static int launch_update(kh_engine *e, const KhUpdateArgs &u, hipStream_t st) {
    const KhSweepArgs p = sweep_args(e, false);
    const KhExchange ex = exchange_args(e, u.internal_exchange != 0);
    if (u.internal_exchange) KH_HIP(hipMemsetAsync(e->d_slots, 0, e->slots_bytes, st));
    if (u.internal_exchange && ex.world > 1) KH_HIP(hipMemsetAsync(e->d_wait_ticks, 0, 2 * sizeof(unsigned long long), st));
    // One launch per interval (sharded sweep): every launch re-stages its operator
    // tiles, so the q2 kernels (5 tiles, 320 KiB per objective) lose to the
    // plain tile kernel (2 tiles) there -- measured 39 vs ~20 us per interval.
    const bool stepwise = !u.internal_exchange;
    int rc = KH_OK;
    const bool whole = !stepwise && u.n_begin == 0 && u.n_end == e->nt - 1;
    e->last_update_grid = 0;
    if (e->ens && whole) {
        int ncg = e->ens_ncg, G = e->ens_G;
        if (e->reduced_G > 0)  // fewer, wider workgroups (kh_set_update_workgroups)
            while (G > e->reduced_G && ncg < KH_ENS_MAXCG) ncg *= 2, G = (e->K + 2 * ncg - 1) / (2 * ncg);
        KhExchange exe = ex;
        exe.G = G;
        KhEnsArgs en;
        en.H0 = e->ens_H0;
        en.H1 = e->ens_H1;
        en.scale = e->d_ens_scale;
        const dim3 g(G), b(KH_ENS_THREADS);
        e->last_update_grid = G;
        const size_t lds = kh_ens_lds_bytes(ncg);
        const bool so = u.sigma != nullptr;
        // first order, four objectives per workgroup (512 < K <= 1024): the A^2 chain with the update sums on the adjoint
        // side (kh_ens2_forward_update; KH_ENS2=0: the term-by-term kernel, A/B switch and what second order and the other
        // column-group counts run -- with one column group it measured 2 % slower, with 4 and 8 its operands do not fit)
        if (!so && ncg == 2 && e->ens2 && e->d_sq_fw != nullptr && ensure_gen_adj(e)) {
            const dim3 agrid((unsigned)e->K, (unsigned)((e->nt + KH_GEN_ADJ_POINTS - 1) / KH_GEN_ADJ_POINTS));
            kh_gen_adjoint_side<<<agrid, KH_GEN_ADJ_THREADS, 0, st>>>(e->d_ops_bw, u.chi_store, e->d_gen_adj, e->K, e->N, 1, e->nt);
            KH_HIP(hipGetLastError());
            KhUpdateArgs ua = u;
            ua.adj_store = e->d_gen_adj;
            const size_t lds2 = kh_ens2_lds_bytes(ncg);
            rc = ensure_dynamic_lds(e, (const void *)kh_ens2_forward_update<2>, lds2);
            // (the interval's first pass sits between the sums' stores and the first poll: no head start on top -- 16.08 -> 15.75 us)
            KhExchange exe2 = exe;
            if (!e->poll_delay_set) exe2.first_poll_delay = 0;
            if (rc == KH_OK) rc = launch_persistent<kh_ens2_forward_update<2>>(e, g, b, lds2, st, p, en, e->d_sq_fw, ua, exe2);
        } else
#define KH_ENS_UPDATE(NCG)                                                                          \
    (so ? launch_persistent<kh_ens_forward_update<NCG, true>>(e, g, b, lds, st, p, en, u, exe)      \
        : launch_persistent<kh_ens_forward_update<NCG, false>>(e, g, b, lds, st, p, en, u, exe))
        switch (ncg) {
            case 1: rc = KH_ENS_UPDATE(1); break;
            case 2: rc = KH_ENS_UPDATE(2); break;
            case 4: rc = KH_ENS_UPDATE(4); break;
            default: rc = KH_ENS_UPDATE(8); break;
        }
#undef KH_ENS_UPDATE
    } else if ((e->stream || (e->reduced_G > 0 && e->reduced_G < e->grid_update && ex.world == 1)) && whole) {
        // more objectives than co-resident workgroups -- or (kh_set_update_workgroups) fewer workgroups than the
        // register-tile family of this engine would use: G workgroups walk through K / G objectives each
        int G = e->stream ? e->stream_G : e->grid_update;
        if (e->reduced_G > 0 && e->reduced_G < G) G = e->reduced_G;
        KhExchange exs = ex;
        exs.G = G;
        exs.world = 1;
        const dim3 g(G), b(512);
        e->last_update_grid = G;
        const bool so = u.sigma != nullptr;
#define KH_STREAM_UPDATE_N(LT, N64)                                                                                \
    (so ? launch_persistent<kh_stream_forward_update<LT, true, N64>>(e, g, b, 0, st, p, u, exs)                   \
        : launch_persistent<kh_stream_forward_update<LT, false, N64>>(e, g, b, 0, st, p, u, exs))
#define KH_STREAM_UPDATE(LT) (e->N == KH_TILE_N ? KH_STREAM_UPDATE_N(LT, true) : KH_STREAM_UPDATE_N(LT, false))
        switch (e->L) {
            case 1: rc = KH_STREAM_UPDATE(1); break;
            case 2: rc = KH_STREAM_UPDATE(2); break;
            case 3: rc = KH_STREAM_UPDATE(3); break;
            case 4: rc = KH_STREAM_UPDATE(4); break;
            default: return kh_fail(KH_ERR_UNSUPPORTED, "tile kernels handle 1..4 controls");
        }
#undef KH_STREAM_UPDATE
#undef KH_STREAM_UPDATE_N
    } else if (e->kind == KIND_TILE_Q2 && e->quad && !stepwise && u.n_begin == 0 && u.n_end == e->nt - 1) {
        if (u.sigma != nullptr)
            launch_plain<kh_quad_forward_update<true>>(dim3(1), dim3(64), 0, st, p, e->d_sq_fw, u, ex);
        else
            launch_plain<kh_quad_forward_update<false>>(dim3(1), dim3(64), 0, st, p, e->d_sq_fw, u, ex);
    } else if (e->kind == KIND_TILE_Q2 && e->mini && !stepwise && u.n_begin == 0 && u.n_end == e->nt - 1) {
        if (u.sigma != nullptr)
            launch_plain<kh_mini_forward_update<true>>(dim3(1), dim3(64 * e->K), 0, st, p, e->d_sq_fw, u, ex);
        else
            launch_plain<kh_mini_forward_update<false>>(dim3(1), dim3(64 * e->K), 0, st, p, e->d_sq_fw, u, ex);
    } else if (e->kind == KIND_TILE_Q2 && !stepwise) {
        const dim3 g(e->K), b(KH_Q2_THREADS);
        // (KH_Q2_SINGLE=0: the instantiations with the cross-GPU stage on one GPU too -- A/B switch)
        const bool single = ex.world == 1 && e->q2_single;
        if (u.sigma != nullptr && single)
            rc = launch_persistent<kh_q2_forward_update<true, false, true>>(e, g, b, kh_q2_lds_bytes(), st, p, e->d_sq_fw, u, ex);
        else if (u.sigma != nullptr)
            rc = launch_persistent<kh_q2_forward_update<true, false>>(e, g, b, kh_q2_lds_bytes(), st, p, e->d_sq_fw, u, ex);
        else if (u.adj_sign != 0.0) {
            KhExchange exa = ex;
            exa.first_poll_delay = e->adj_poll_delay;
            if (single)  // (the default form on one GPU: an instantiation without the cross-GPU stage)
                rc = launch_persistent<kh_q2_forward_update<false, true, true>>(e, g, b, kh_q2_lds_bytes(), st, p, e->d_sq_fw, u, exa);
            else
                rc = launch_persistent<kh_q2_forward_update<false, true>>(e, g, b, kh_q2_lds_bytes(), st, p, e->d_sq_fw, u, exa);
        } else if (single)
            rc = launch_persistent<kh_q2_forward_update<false, false, true>>(e, g, b, kh_q2_lds_bytes(), st, p, e->d_sq_fw, u, ex);
        else
            rc = launch_persistent<kh_q2_forward_update<false, false>>(e, g, b, kh_q2_lds_bytes(), st, p, e->d_sq_fw, u, ex);
    } else if (e->kind == KIND_COOP && !stepwise) {
        if (e->coop_cols == 2)
            rc = e->coop_ks <= 8 ? launch_coop_update<8, 2>(e, p, u, ex, st) : launch_coop_update<16, 2>(e, p, u, ex, st);
        else if (e->coop_cols == 4)
            rc = e->coop_ks <= 8 ? launch_coop_update<8, 4>(e, p, u, ex, st) : launch_coop_update<16, 4>(e, p, u, ex, st);
        else
            rc = e->coop_ks <= 8 ? launch_coop_update<8, 16>(e, p, u, ex, st) : launch_coop_update<16, 16>(e, p, u, ex, st);
    } else if (e->kind == KIND_TILEN && !stepwise) {
        const dim3 g(e->K), b(KH_TN_THREADS);
        const size_t lds = kh_tn_lds_bytes();
        const bool so = u.sigma != nullptr;
        // first order with the control operators NOT in registers (several controls, or N > 96): the update sums on the
        // adjoint side, H_lk^+ chi_k for the whole store in front of the sweep (kh_generic.h, kh_gen_adjoint_side) instead
        // of L streamed control products per interval (N = 100, L = 6: 60 of the update sweep's 106 us per interval)
        KhUpdateArgs ut = u;
        if (!so && !e->tn_h1reg && ensure_gen_adj(e)) {
            const dim3 agrid((unsigned)(e->K * e->L), (unsigned)((e->nt + KH_GEN_ADJ_POINTS - 1) / KH_GEN_ADJ_POINTS));
            kh_gen_adjoint_side<<<agrid, KH_GEN_ADJ_THREADS, 0, st>>>(e->d_ops_bw, u.chi_store, e->d_gen_adj, e->K, e->N, e->L, e->nt);
            KH_HIP(hipGetLastError());
            ut.adj_store = e->d_gen_adj;
        }
        const KhUpdateArgs &u = ut;
        // (several waves poll side by side: a later first poll, as for the tile64x kernels -- N = 100, L = 6: 68.8 -> 64.9 us per
        // interval, N = 81, L = 4: 38.2 -> 35.5; KH_POLL_DELAY overrides)
        KhExchange ext = ex;
        if (!e->poll_delay_set && e->L >= 2) ext.first_poll_delay = e->L == 2 ? 32 : 64;
        const KhExchange &ex = ext;
#define KH_TN_UPDATE(EP, HR)                                                                                                  \
    (so ? launch_persistent<kh_tn_forward_update<EP, true, HR>>(e, g, b, lds, st, p, (const cplx *const *)e->d_tn_fw, u, ex) \
        : launch_persistent<kh_tn_forward_update<EP, false, HR>>(e, g, b, lds, st, p, (const cplx *const *)e->d_tn_fw, u, ex))
        if (e->N <= 80)
            rc = e->tn_h1reg ? KH_TN_UPDATE(20, true) : KH_TN_UPDATE(20, false);
        else if (e->N <= 96)
            rc = e->tn_h1reg ? KH_TN_UPDATE(24, true) : KH_TN_UPDATE(24, false);
        else
            rc = e->N <= 112 ? KH_TN_UPDATE(28, false) : KH_TN_UPDATE(32, false);
#undef KH_TN_UPDATE
    } else if (e->kind == KIND_ELL && !stepwise) {
        const dim3 g(e->K);
        const size_t lds = kh_ell_lds_bytes(e->ell_stream);
        const bool so = u.sigma != nullptr;
#define KH_ELL_UPDATE_SO(T, R, EM, SO)                                                                                            \
    (ensure_dynamic_lds(e, (const void *)kh_ell_forward_update<T, R, EM, SO>, lds) != KH_OK                                        \
         ? KH_ERR_HIP                                                                                                              \
         : launch_persistent<kh_ell_forward_update<T, R, EM, SO>>(e, g, dim3(T), lds, st, p, (const KhEll *)e->d_ell_fw,            \
                                                                  (const int *)e->d_ell_off, (const cplx *)e->d_ell_vals, u, ex,    \
                                                                  (cplx *)nullptr, 0LL))
#define KH_ELL_UPDATE(T, R, EM) (so ? KH_ELL_UPDATE_SO(T, R, EM, true) : KH_ELL_UPDATE_SO(T, R, EM, false))
        if (e->ell_stream) {
            rc = ensure_dynamic_lds(e, so ? (const void *)kh_ell_forward_update<512, KH_ELLS_RPL, 4, true, true>
                                          : (const void *)kh_ell_forward_update<512, KH_ELLS_RPL, 4, false, true>, lds);
            if (rc == KH_OK)
                rc = so ? launch_persistent<kh_ell_forward_update<512, KH_ELLS_RPL, 4, true, true>>(
                              e, g, dim3(512), lds, st, p, (const KhEll *)e->d_ell_fw, (const int *)e->d_ell_off,
                              (const cplx *)e->d_ell_vals, u, ex, e->d_ell_scratch, e->ell_scratch_stride)
                        : launch_persistent<kh_ell_forward_update<512, KH_ELLS_RPL, 4, false, true>>(
                              e, g, dim3(512), lds, st, p, (const KhEll *)e->d_ell_fw, (const int *)e->d_ell_off,
                              (const cplx *)e->d_ell_vals, u, ex, e->d_ell_scratch, e->ell_scratch_stride);
        } else if (e->N <= 512)
            rc = e->ell_E <= 8 ? KH_ELL_UPDATE(512, 1, 8) : e->ell_E <= 12 ? KH_ELL_UPDATE(512, 1, 12)
                 : e->ell_E <= 16 ? KH_ELL_UPDATE(512, 1, 16) : e->ell_E <= 24 ? KH_ELL_UPDATE(512, 1, 24) : KH_ELL_UPDATE(512, 1, 32);
        else if (e->N <= 768 && e->ell_E <= 16)
            rc = e->ell_E <= 8 ? KH_ELL_UPDATE(768, 1, 8) : e->ell_E <= 12 ? KH_ELL_UPDATE(768, 1, 12) : KH_ELL_UPDATE(768, 1, 16);
        else if (e->N > 1024)
            rc = e->N <= 1536 ? KH_ELL_UPDATE(512, 3, 8) : KH_ELL_UPDATE(512, 4, 8);
        else if (e->ell_E <= 8)
            rc = KH_ELL_UPDATE(1024, 1, 8);
        else
            rc = e->ell_E <= 12 ? KH_ELL_UPDATE(512, 2, 12) : KH_ELL_UPDATE(512, 2, 16);
#undef KH_ELL_UPDATE
#undef KH_ELL_UPDATE_SO
    } else if (e->kind != KIND_GENERIC && e->kind != KIND_COOP && e->kind != KIND_ELL && e->kind != KIND_TILEN) {
        const bool rpt2 = e->kind == KIND_TILE_RPT2;
        switch (e->L) {
            case 1: rc = rpt2 ? launch_tile_update<2, 1>(e, p, u, ex, st) : launch_tile_update<1, 1>(e, p, u, ex, st); break;
            case 2: rc = launch_tile_update<1, 2>(e, p, u, ex, st); break;
            case 3: rc = launch_tile_update<1, 3>(e, p, u, ex, st); break;
            case 4: rc = launch_tile_update<1, 4>(e, p, u, ex, st); break;
            default: return kh_fail(KH_ERR_UNSUPPORTED, "tile kernels handle 1..4 controls");
        }
    } else if (e->tx_update && whole && u.sigma == nullptr && e->reduced_G == 0 && e->grid_update >= e->K && ensure_gen_adj(e)) {
        // five to eight controls, N <= 64, one resident workgroup per objective, first order: the register-tile form with
        // streamed operators (kh_tile64x.h); V_lk = H_lk^+ chi_k for the whole store in front of the sweep as below
        const dim3 agrid((unsigned)(e->K * e->L), (unsigned)((e->nt + KH_GEN_ADJ_POINTS - 1) / KH_GEN_ADJ_POINTS));
        kh_gen_adjoint_side<<<agrid, KH_GEN_ADJ_THREADS, 0, st>>>(e->d_ops_bw, u.chi_store, e->d_gen_adj, e->K, e->N, e->L, e->nt);
        KH_HIP(hipGetLastError());
        KhUpdateArgs ux = u;
        ux.adj_store = e->d_gen_adj;
        KhExchange exx = ex;
        exx.G = e->K;
        // (five to eight waves poll side by side: their first poll later than the register-tile kernels' -- measured best at
        // K = 256, N = 64: 32 / 44 / 64 / 64 for L = 5 / 6 / 7 / 8, profiles/r06/ab_tile64x.txt; KH_POLL_DELAY overrides)
        if (!e->poll_delay_set) exx.first_poll_delay = e->L <= 5 ? 32 : e->L == 6 ? 44 : 64;
        e->last_update_grid = e->K;
#define KH_TX_UPDATE(LT)                                                                                              \
    do {                                                                                                              \
        rc = ensure_dynamic_lds(e, (const void *)kh_tx_forward_update<LT>, kh_tx_lds_bytes());                        \
        if (rc == KH_OK)                                                                                              \
            rc = launch_persistent<kh_tx_forward_update<LT>>(e, dim3(e->K), dim3(KH_TX_THREADS), kh_tx_lds_bytes(), st, p, \
                                                             (const cplx *const *)e->d_tx_fw, ux, exx);                \
    } while (0)
        switch (e->L) {
            case 5: KH_TX_UPDATE(5); break;
            case 6: KH_TX_UPDATE(6); break;
            case 7: KH_TX_UPDATE(7); break;
            default: KH_TX_UPDATE(8); break;
        }
#undef KH_TX_UPDATE
    } else {
        if (!e->gen_fits)
            return kh_fail(KH_ERR_UNSUPPORTED, "N=%d: this form of the update sweep needs the generic kernels, whose vectors do not fit LDS", e->N);
        const size_t lds = kh_gen_lds_bytes(e->N, e->d_csr_fw == nullptr);
        rc = ensure_dynamic_lds(e, (const void *)kh_gen_forward_update, lds);
        ensure_gen_scratch(e, e->grid_update);
        KhSweepArgs pg = p;
        pg.gen_scratch = e->d_gen_scratch;
        pg.gen_scratch_wgs = e->gen_scratch_wgs;
        // first order, dense operators: V_lk = H_lk^+ chi_k for the whole store in front of the sweep (kh_generic.h,
        // kh_gen_adjoint_side) -- at the sweep's first launch (the single launch, or kh_update_begin's); the launches
        // of a stepwise sweep that follow reuse it
        KhUpdateArgs ug = u;
        if (rc == KH_OK) {
            const bool first_launch = u.internal_exchange || (u.n_dev == nullptr && u.n_begin == 0 && u.n_end == 0);
            if (first_launch) {
                e->gen_adj_ready = false;
                if (u.sigma == nullptr && e->d_csr_fw == nullptr && ensure_gen_adj(e)) {
                    const dim3 grid((unsigned)(e->K * e->L), (unsigned)((e->nt + KH_GEN_ADJ_POINTS - 1) / KH_GEN_ADJ_POINTS));
                    kh_gen_adjoint_side<<<grid, KH_GEN_ADJ_THREADS, 0, st>>>(e->d_ops_bw, u.chi_store, e->d_gen_adj, e->K, e->N,
                                                                              e->L, e->nt);
                    KH_HIP(hipGetLastError());
                    e->gen_adj_ready = true;
                }
            }
            if (e->gen_adj_ready && u.sigma == nullptr) ug.adj_store = e->d_gen_adj;
        }
        if (rc == KH_OK && !ug.internal_exchange)
            launch_plain<kh_gen_forward_update>(dim3(e->grid_update), dim3(KH_GEN_THREADS), lds, st, pg, ug, ex);
        else if (rc == KH_OK)
            rc = launch_persistent<kh_gen_forward_update>(e, dim3(e->grid_update), dim3(KH_GEN_THREADS), lds, st, pg, ug, ex);
    }
    if (rc != KH_OK) return rc;
    KH_HIP(hipGetLastError());
    return KH_OK;
}

static KhUpdateArgs update_args(kh_engine *e, const kh_cdouble *chi_store, const double *chi_norms,
                                const double *guess, const double *shape, const double *lambda, double *opt,
                                double *g_a) {
    KhUpdateArgs u;
    u.mu_re = e->is_super ? 0.0 : 1.0;  // mu.py:130-134
    u.mu_im = e->is_super ? 1.0 : 0.0;
    u.chi_store = (const cplx *)chi_store;
    u.chi_norms = chi_norms;
    u.phi = e->d_phi;
    u.guess = guess;
    u.shape = shape;
    u.lambda = lambda;
    u.opt = opt;
    u.g_a = g_a;
    u.wg_partial = e->d_wg_partial;
    u.D_in = nullptr;
    u.n_dev = nullptr;
    u.fw_prev = e->so_fw_prev;
    u.fw_store = e->so_fw_store;
    u.sigma = e->so_sigma;
    u.n_begin = 0;
    u.n_end = e->nt - 1;
    u.internal_exchange = 1;
    u.adj_sign = e->adj_sign;
    u.adj_store = nullptr;
    return u;
}

extern "C" int kh_forward_update(kh_engine *e, const kh_cdouble *chi_store_dev, const double *chi_norms_dev,
                                 const kh_cdouble *init_dev, const double *guess_dev, const double *shape_dev,
                                 const double *lambda_dev, double *opt_dev, kh_cdouble *psi_T_dev, double *g_a_dev,
                                 void *stream) {
    if (e == nullptr || chi_store_dev == nullptr || chi_norms_dev == nullptr || init_dev == nullptr ||
        guess_dev == nullptr || shape_dev == nullptr || lambda_dev == nullptr || opt_dev == nullptr ||
        psi_T_dev == nullptr || g_a_dev == nullptr)
        return kh_fail(KH_ERR_INVALID, "null argument");
    if (e->L < 1) return kh_fail(KH_ERR_INVALID, "no controls to update");
    if (opt_dev == guess_dev) return kh_fail(KH_ERR_INVALID, "opt_dev must not alias guess_dev");
    hipStream_t st = (hipStream_t)stream;
    if (e->stepwise_only && !e->stream && !e->ens) {
        // one launch per interval; on one GPU the "all-reduced" sums are the local ones (kh_reduce_partials has
        // summed the workgroups' pieces in a fixed order)
        KH_HIP(hipMemsetAsync(e->d_stats, 0, sizeof(double) * 4, st));
        int rc = kh_update_begin(e, chi_store_dev, chi_norms_dev, init_dev, guess_dev, opt_dev, g_a_dev,
                                 e->d_step_partial, stream);
        for (int n = 0; rc == KH_OK && n < e->nt - 1; ++n)
            rc = kh_update_step(e, n, e->d_step_partial, chi_store_dev, chi_norms_dev, shape_dev, lambda_dev, opt_dev,
                                g_a_dev, e->d_step_partial, stream);
        if (rc == KH_OK) rc = kh_update_end(e, psi_T_dev, stream);
        e->last_intervals = e->nt - 1;
        e->last_wgs = e->grid_update;
        return rc;
    }
#ifdef KH_TIMING
    KH_HIP(hipMemsetAsync(e->d_stats, 0, sizeof(double) * 68, st));
#else
    KH_HIP(hipMemsetAsync(e->d_stats, 0, sizeof(double) * 4, st));
#endif
    KH_HIP(hipMemcpyAsync(e->d_phi, init_dev, sizeof(cplx) * (size_t)e->K * e->N, hipMemcpyDeviceToDevice, st));
    KhUpdateArgs u =
        update_args(e, chi_store_dev, chi_norms_dev, guess_dev, shape_dev, lambda_dev, opt_dev, g_a_dev);
    int rc = launch_update(e, u, st);
    if (rc != KH_OK) return rc;
    if (e->p2p_ready) {
        e->p2p_epoch_base += (unsigned int)e->nt;  // every rank advances identically
        e->p2p_sweeps += 1;
    }
    KH_HIP(hipMemcpyAsync(psi_T_dev, e->d_phi, sizeof(cplx) * (size_t)e->K * e->N, hipMemcpyDeviceToDevice, st));
    e->last_intervals = e->nt - 1;
    e->last_wgs = e->last_update_grid > 0 ? e->last_update_grid : e->grid_update;
    return KH_OK;
}

// Fewer workgroups for the single-launch update sweep: what a caller asks for after KH_ERR_TIMEOUT (co-tenants hold
// compute units, so the sweep's workgroups were not all resident at once) before it gives up on the single launch.
extern "C" int kh_set_update_workgroups(kh_engine *e, int32_t max_workgroups, int32_t *chosen) {
    if (e == nullptr || max_workgroups < 0) return kh_fail(KH_ERR_INVALID, "bad argument");
    if (chosen != nullptr) *chosen = e->ens ? e->ens_G : (e->stream ? e->stream_G : e->grid_update);
    if (max_workgroups == 0) {
        e->reduced_G = 0;
        return KH_OK;
    }
    if (e->p2p_ready) return kh_fail(KH_ERR_UNSUPPORTED, "sharded sweeps keep their grid (all ranks must agree on the form)");
    int G = 0;
    if (e->ens) {
        const int widest = (e->K + 2 * KH_ENS_MAXCG - 1) / (2 * KH_ENS_MAXCG);
        if (max_workgroups < widest)
            return kh_fail(KH_ERR_UNSUPPORTED, "the ensemble kernel needs at least %d workgroups for %d objectives", widest, e->K);
        int ncg = e->ens_ncg;
        G = e->ens_G;
        while (G > max_workgroups && ncg < KH_ENS_MAXCG) ncg *= 2, G = (e->K + 2 * ncg - 1) / (2 * ncg);
        const void *forms[2] = {nullptr, nullptr};
        switch (ncg) {
            case 1: forms[0] = (const void *)kh_ens_forward_update<1, false>, forms[1] = (const void *)kh_ens_forward_update<1, true>; break;
            case 2: forms[0] = (const void *)kh_ens_forward_update<2, false>, forms[1] = (const void *)kh_ens_forward_update<2, true>; break;
            case 4: forms[0] = (const void *)kh_ens_forward_update<4, false>, forms[1] = (const void *)kh_ens_forward_update<4, true>; break;
            default: forms[0] = (const void *)kh_ens_forward_update<8, false>, forms[1] = (const void *)kh_ens_forward_update<8, true>; break;
        }
        for (const void *f : forms) {
            const int rc = ensure_dynamic_lds(e, f, kh_ens_lds_bytes(ncg));
            if (rc != KH_OK) return rc;
        }
    } else {
        const bool tile_family = (e->kind == KIND_TILE_Q2 && !e->mini) || e->kind == KIND_TILE_RPT1 || e->kind == KIND_TILE_RPT2;
        if (!tile_family || e->d_csr_fw != nullptr || e->N > KH_TILE_N || e->L < 1 || e->L > 4)
            return kh_fail(KH_ERR_UNSUPPORTED, "only the register-tile families (N <= 64, 1..4 controls) have a form with fewer workgroups");
        const int fewest = (e->K + KH_STREAM_MMAX - 1) / KH_STREAM_MMAX;
        if (max_workgroups < fewest)
            return kh_fail(KH_ERR_UNSUPPORTED, "%d objectives need at least %d workgroups (%d per workgroup)", e->K, fewest, KH_STREAM_MMAX);
        G = e->stream ? e->stream_G : e->grid_update;
        if (max_workgroups < G) G = max_workgroups;
    }
    e->reduced_G = max_workgroups;
    if (chosen != nullptr) *chosen = G;
    return KH_OK;
}

extern "C" int kh_set_second_order(kh_engine *e, const kh_cdouble *fw_prev_dev, kh_cdouble *fw_store_dev,
                                   const double *sigma_dev) {
    if (e == nullptr) return kh_fail(KH_ERR_INVALID, "null engine");
    const int given = (fw_prev_dev != nullptr) + (fw_store_dev != nullptr) + (sigma_dev != nullptr);
    if (given != 0 && given != 3)
        return kh_fail(KH_ERR_INVALID, "fw_prev, fw_store and sigma must be given together (or all NULL)");
    if (fw_prev_dev != nullptr && (const void *)fw_prev_dev == (const void *)fw_store_dev)
        return kh_fail(KH_ERR_INVALID, "fw_store must not alias fw_prev");
    e->so_fw_prev = (const cplx *)fw_prev_dev;
    e->so_fw_store = (cplx *)fw_store_dev;
    e->so_sigma = sigma_dev;
    return KH_OK;
}

extern "C" int kh_update_begin(kh_engine *e, const kh_cdouble *chi_store_dev, const double *chi_norms_dev,
                               const kh_cdouble *init_dev, const double *guess_dev, double *opt_dev,
                               double *g_a_dev, double *partial_dev, void *stream) {
    if (e == nullptr || chi_store_dev == nullptr || chi_norms_dev == nullptr || init_dev == nullptr ||
        guess_dev == nullptr || opt_dev == nullptr || g_a_dev == nullptr || partial_dev == nullptr)
        return kh_fail(KH_ERR_INVALID, "null argument");
    if (e->L < 1) return kh_fail(KH_ERR_INVALID, "no controls to update");
    hipStream_t st = (hipStream_t)stream;
    e->guess_dev = guess_dev;
    KH_HIP(hipMemsetAsync(e->d_stats, 0, sizeof(double) * 4, st));
    KH_HIP(hipMemcpyAsync(e->d_phi, init_dev, sizeof(cplx) * (size_t)e->K * e->N, hipMemcpyDeviceToDevice, st));
    KH_HIP(hipMemcpyAsync(opt_dev, guess_dev, sizeof(double) * (size_t)e->L * (e->nt - 1),
                          hipMemcpyDeviceToDevice, st));
    KH_HIP(hipMemsetAsync(g_a_dev, 0, sizeof(double) * e->L, st));
    KhUpdateArgs u = update_args(e, chi_store_dev, chi_norms_dev, guess_dev, nullptr, nullptr, opt_dev, g_a_dev);
    u.internal_exchange = 0;
    u.n_begin = u.n_end = 0;
    int rc = launch_update(e, u, st);
    if (rc != KH_OK) return rc;
    kh_reduce_partials<<<1, 64, 0, st>>>(e->d_wg_partial, e->grid_update, e->L, partial_dev, nullptr);
    KH_HIP(hipGetLastError());
    return KH_OK;
}

extern "C" int kh_update_step(kh_engine *e, int32_t n, const double *D_dev, const kh_cdouble *chi_store_dev,
                              const double *chi_norms_dev, const double *shape_dev, const double *lambda_dev,
                              double *opt_dev, double *g_a_dev, double *partial_dev, void *stream) {
    if (e == nullptr || D_dev == nullptr || chi_store_dev == nullptr || chi_norms_dev == nullptr ||
        shape_dev == nullptr || lambda_dev == nullptr || opt_dev == nullptr || g_a_dev == nullptr ||
        partial_dev == nullptr)
        return kh_fail(KH_ERR_INVALID, "null argument");
    if (e->guess_dev == nullptr) return kh_fail(KH_ERR_INVALID, "kh_update_begin was not called");
    if (n < 0 || n >= e->nt - 1) return kh_fail(KH_ERR_INVALID, "interval %d out of range", n);
    hipStream_t st = (hipStream_t)stream;
    KhUpdateArgs u =
        update_args(e, chi_store_dev, chi_norms_dev, e->guess_dev, shape_dev, lambda_dev, opt_dev, g_a_dev);
    u.internal_exchange = 0;
    u.D_in = D_dev;
    u.n_begin = n;
    u.n_end = n + 1;
    int rc = launch_update(e, u, st);
    if (rc != KH_OK) return rc;
    if (n + 1 < e->nt - 1) {
        kh_reduce_partials<<<1, 64, 0, st>>>(e->d_wg_partial, e->grid_update, e->L, partial_dev, nullptr);
        KH_HIP(hipGetLastError());
    }
    return KH_OK;
}

extern "C" int kh_update_step_dev(kh_engine *e, int32_t *n_dev, const double *D_dev,
                                  const kh_cdouble *chi_store_dev, const double *chi_norms_dev,
                                  const double *shape_dev, const double *lambda_dev, double *opt_dev,
                                  double *g_a_dev, double *partial_dev, void *stream) {
    if (e == nullptr || n_dev == nullptr || D_dev == nullptr || chi_store_dev == nullptr ||
        chi_norms_dev == nullptr || shape_dev == nullptr || lambda_dev == nullptr || opt_dev == nullptr ||
        g_a_dev == nullptr || partial_dev == nullptr)
        return kh_fail(KH_ERR_INVALID, "null argument");
    if (e->guess_dev == nullptr) return kh_fail(KH_ERR_INVALID, "kh_update_begin was not called");
    hipStream_t st = (hipStream_t)stream;
    KhUpdateArgs u =
        update_args(e, chi_store_dev, chi_norms_dev, e->guess_dev, shape_dev, lambda_dev, opt_dev, g_a_dev);
    u.internal_exchange = 0;
    u.D_in = D_dev;
    u.n_dev = n_dev;
    u.n_begin = 0;  // overridden on the device
    u.n_end = 1;
    int rc = launch_update(e, u, st);
    if (rc != KH_OK) return rc;
    kh_reduce_partials<<<1, 64, 0, st>>>(e->d_wg_partial, e->grid_update, e->L, partial_dev, n_dev);
    KH_HIP(hipGetLastError());
    return KH_OK;
}

extern "C" int kh_update_end(kh_engine *e, kh_cdouble *psi_T_dev, void *stream) {
    if (e == nullptr || psi_T_dev == nullptr) return kh_fail(KH_ERR_INVALID, "null argument");
    KH_HIP(hipMemcpyAsync(psi_T_dev, e->d_phi, sizeof(cplx) * (size_t)e->K * e->N, hipMemcpyDeviceToDevice,
                          (hipStream_t)stream));
    e->guess_dev = nullptr;
    return KH_OK;
}


// ---------------------------------------------------------------------------
// cross-GPU exchange windows (sharded objectives, one rank per GPU)
// ---------------------------------------------------------------------------

// in-kernel ping-pong over the windows: every rank publishes (rank + round) and
// must read the same total from its own window
__global__ void kh_p2p_selftest_kernel(KhExchange ex, int L, int rounds, unsigned int epoch0, int *result) {
    const int lane = threadIdx.x;
    int ok_all = 1;
    long long t_first = 0;
    for (int r = 0; r < rounds; ++r) {
        if (r == 1) t_first = wall_clock64();  // (the first round absorbs the ranks' launch skew)
        double vals[KH_MAX_L], out[KH_MAX_L];
        for (int l = 0; l < KH_MAX_L; ++l) vals[l] = (double)(ex.rank + 1) * (l + 1) + 0.25 * r;
        const unsigned int epoch = epoch0 + (unsigned)r + 1u;
        kh_p2p_publish(ex, r & 1, L, lane, vals, epoch);
        if (!kh_p2p_gather<KH_MAX_L>(ex, r & 1, L, epoch, lane, out)) {
            ok_all = 0;
            break;
        }
        for (int l = 0; l < L; ++l) {
            const double want = (double)(ex.world * (ex.world + 1) / 2) * (l + 1) + 0.25 * r * ex.world;
            if (out[l] != want) ok_all = 0;
        }
    }
    if (lane == 0) *result = ok_all;
    if (lane == 0 && ex.wait_ticks != nullptr && rounds > 1) {  // publish -> all ranks' values read back, per round
        ex.wait_ticks[2] = (unsigned long long)(wall_clock64() - t_first);
        ex.wait_ticks[3] = (unsigned long long)(rounds - 1);
    }
}

extern "C" int kh_p2p_create_window(kh_engine *e, int32_t world, int32_t rank, unsigned char *ipc_handle_out) {
    if (e == nullptr || ipc_handle_out == nullptr) return kh_fail(KH_ERR_INVALID, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return kh_fail(KH_ERR_INVALID, "bad world/rank %d/%d", rank, world);
    const int Lx = e->L > 0 ? e->L : 1;
    if (world * Lx * 2 > 64 || Lx > KH_MAX_L)
        return kh_fail(KH_ERR_UNSUPPORTED, "world * L = %d exceeds the 32 exchange lanes (or more than %d controls)", world * Lx, KH_MAX_L);
    if (e->stepwise_only && !e->ens)  // (the caller falls back to kh_update_step + an all-reduce per interval)
        return kh_fail(KH_ERR_UNSUPPORTED, "%d objectives per GPU are not co-resident: no in-kernel exchange", e->K);
    if (e->p2p_window != nullptr) return kh_fail(KH_ERR_INVALID, "window already created");
    e->p2p_world = world;
    e->p2p_rank = rank;
    e->p2p_window_bytes = sizeof(kh_u64) * 2 * (size_t)world * Lx * 2;
    if (e->p2p_window_bytes < 4096) e->p2p_window_bytes = 4096;
    void *ptr = nullptr;
    // fine-grained (uncached, system-coherent) device memory for cross-GPU visibility
    hipError_t err = hipExtMallocWithFlags(&ptr, e->p2p_window_bytes, hipDeviceMallocFinegrained);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        KH_HIP(hipMalloc(&ptr, e->p2p_window_bytes));
    }
    e->p2p_window = (kh_u64 *)ptr;
    KH_HIP(hipMemset(ptr, 0, e->p2p_window_bytes));
    hipIpcMemHandle_t h;
    KH_HIP(hipIpcGetMemHandle(&h, ptr));
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle larger than the ABI's 64 bytes");
    memset(ipc_handle_out, 0, 64);
    memcpy(ipc_handle_out, &h, sizeof(h));
    KH_HIP(hipDeviceSynchronize());
    return KH_OK;
}

extern "C" int kh_p2p_open_peers(kh_engine *e, const unsigned char *all_handles) {
    if (e == nullptr || all_handles == nullptr) return kh_fail(KH_ERR_INVALID, "null argument");
    if (e->p2p_window == nullptr) return kh_fail(KH_ERR_INVALID, "kh_p2p_create_window was not called");
    std::vector<kh_u64 *> peers(e->p2p_world, nullptr);
    for (int r = 0; r < e->p2p_world; ++r) {
        if (r == e->p2p_rank) {
            peers[r] = e->p2p_window;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, all_handles + (size_t)r * 64, sizeof(h));
        void *ptr = nullptr;
        KH_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        e->p2p_opened.push_back(ptr);
        peers[r] = (kh_u64 *)ptr;
    }
    KH_HIP(hipMalloc((void **)&e->d_p2p_peers, sizeof(kh_u64 *) * e->p2p_world));
    KH_HIP(hipMemcpy((void *)e->d_p2p_peers, peers.data(), sizeof(kh_u64 *) * e->p2p_world, hipMemcpyHostToDevice));
    return KH_OK;
}

// Collective over all ranks (each calls it at the same point): returns KH_OK when
// `rounds` in-kernel exchanges over the windows produced the expected totals on
// THIS rank; the caller combines the verdicts of all ranks.  Marks the engine
// ready for the cross-GPU update sweep on success.
extern "C" int kh_p2p_selftest(kh_engine *e, int32_t rounds, void *stream) {
    if (e == nullptr) return kh_fail(KH_ERR_INVALID, "null engine");
    if (e->d_p2p_peers == nullptr) return kh_fail(KH_ERR_INVALID, "kh_p2p_open_peers was not called");
    hipStream_t st = (hipStream_t)stream;
    KhExchange ex;
    memset(&ex, 0, sizeof(ex));
    ex.abort_flag = e->d_abort;
    ex.timeout_ticks = 20000000LL;  // 0.2 s
    ex.peer_windows = e->d_p2p_peers;
    ex.my_window = e->p2p_window;
    ex.world = e->p2p_world;
    ex.rank = e->p2p_rank;
    ex.fail_at = -1;
    ex.wait_ticks = e->d_wait_ticks;
    int *d_res = nullptr;
    KH_HIP(hipMalloc(&d_res, sizeof(int)));
    KH_HIP(hipMemsetAsync(d_res, 0, sizeof(int), st));
    const int Lx = e->L > 0 ? e->L : 1;
    kh_p2p_selftest_kernel<<<1, 64, 0, st>>>(ex, Lx, rounds, e->p2p_epoch_base, d_res);
    KH_HIP(hipGetLastError());
    e->p2p_epoch_base += (unsigned int)rounds + 1u;
    int res = 0;
    KH_HIP(hipMemcpyAsync(&res, d_res, sizeof(int), hipMemcpyDeviceToHost, st));
    KH_HIP(hipStreamSynchronize(st));
    (void)hipFree(d_res);
    unsigned int flag = 0;
    KH_HIP(hipMemcpy(&flag, e->d_abort, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag != 0) KH_HIP(hipMemset(e->d_abort, 0, sizeof(flag)));
    if (res != 1 || flag != 0) {
        e->p2p_ready = false;
        return kh_fail(KH_ERR_TIMEOUT, "cross-GPU exchange self-test failed on rank %d", e->p2p_rank);
    }
    e->p2p_ready = true;
    return KH_OK;
}

extern "C" int kh_p2p_stats(kh_engine *e, double out[4]) {
    if (e == nullptr || out == nullptr) return kh_fail(KH_ERR_INVALID, "null argument");
    unsigned long long t[4] = {0, 0, 0, 0};
    KH_HIP(hipMemcpy(t, e->d_wait_ticks, sizeof(t), hipMemcpyDeviceToHost));
    const double n = e->last_intervals > 0 ? e->last_intervals : 1.0;
    out[0] = (double)t[0] * 0.01 / n;                       // 100 MHz ticks -> us
    out[1] = (double)t[1] * 0.01 / n;
    out[2] = t[3] > 0 ? (double)t[2] * 0.01 / (double)t[3] : 0.0;
    out[3] = (double)e->p2p_world;
    return KH_OK;
}

extern "C" int kh_p2p_disable(kh_engine *e) {
    if (e == nullptr) return kh_fail(KH_ERR_INVALID, "null engine");
    e->p2p_ready = false;
    return KH_OK;
}

extern "C" int kh_tau(kh_engine *e, const kh_cdouble *targets_dev, const kh_cdouble *psi_T_dev,
                      kh_cdouble *tau_dev, void *stream) {
    if (e == nullptr || targets_dev == nullptr || psi_T_dev == nullptr || tau_dev == nullptr)
        return kh_fail(KH_ERR_INVALID, "null argument");
    const int waves_per_block = 4;
    const int blocks = (e->K + waves_per_block - 1) / waves_per_block;
    kh_tau_kernel<<<blocks, 64 * waves_per_block, 0, (hipStream_t)stream>>>(
        (const cplx *)targets_dev, (const cplx *)psi_T_dev, (cplx *)tau_dev, e->K, e->N);
    KH_HIP(hipGetLastError());
    return KH_OK;
}

extern "C" int kh_chi_boundary(kh_engine *e, const kh_cdouble *targets_dev, const kh_cdouble *psi_T_dev,
                               const kh_cdouble *c_dev, const kh_cdouble *d_dev, kh_cdouble *chi_T_dev,
                               double *chi_norms_dev, void *stream) {
    if (e == nullptr || targets_dev == nullptr || psi_T_dev == nullptr || c_dev == nullptr || d_dev == nullptr ||
        chi_T_dev == nullptr || chi_norms_dev == nullptr)
        return kh_fail(KH_ERR_INVALID, "null argument");
    const int waves_per_block = 4;
    const int blocks = (e->K + waves_per_block - 1) / waves_per_block;
    kh_chi_kernel<<<blocks, 64 * waves_per_block, 0, (hipStream_t)stream>>>(
        (const cplx *)targets_dev, (const cplx *)psi_T_dev, (const cplx *)c_dev, (const cplx *)d_dev,
        (cplx *)chi_T_dev, chi_norms_dev, e->K, e->N);
    KH_HIP(hipGetLastError());
    return KH_OK;
}

extern "C" int kh_check(kh_engine *e) {
    if (e == nullptr) return kh_fail(KH_ERR_INVALID, "null engine");
    unsigned int flag = 0;
    KH_HIP(hipMemcpy(&flag, e->d_abort, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag != 0) {
        KH_HIP(hipMemset(e->d_abort, 0, sizeof(flag)));
        return kh_fail(KH_ERR_TIMEOUT, "in-kernel exchange timed out: outputs of the last update sweep are invalid");
    }
    return KH_OK;
}

extern "C" int kh_series_tables(int32_t real_spectrum, double tol, double *theta, double *ratios) {
    if (theta == nullptr || ratios == nullptr) return kh_fail(KH_ERR_INVALID, "null argument");
    static_assert(KH_MAX_DEGREE == 64 && KH_RATIO_STRIDE == 65, "documented table sizes");
    if (!(tol > 0.0)) tol = ldexp(1.0, -53);
    std::vector<double> c0(KH_MAX_DEGREE + 1), rows((size_t)(KH_MAX_DEGREE + 1) * KH_Q2_ROWS * 2);
    if (real_spectrum) {
        kh_build_real_spectrum_rows(tol, theta, c0.data(), rows.data(), ratios);
    } else {
        kh_build_degree_table(tol, theta);
        kh_build_taylor_rows(c0.data(), rows.data(), ratios);
    }
    return KH_OK;
}

extern "C" int kh_series_tables_defect(double tol, double theta_cap, double defect, double *theta, double *ratios) {
    if (theta == nullptr || ratios == nullptr) return kh_fail(KH_ERR_INVALID, "null argument");
    if (!(theta_cap > 0.0) || theta_cap > 8.0 || !(defect >= 0.0))
        return kh_fail(KH_ERR_INVALID, "theta_cap must be in (0, 8], defect >= 0");
    if (!(tol > 0.0)) tol = ldexp(1.0, -53);
    std::vector<double> c0(KH_MAX_DEGREE + 1), rows((size_t)(KH_MAX_DEGREE + 1) * KH_Q2_ROWS * 2);
    kh_build_real_spectrum_rows(tol, theta, c0.data(), rows.data(), ratios, theta_cap, defect);
    return KH_OK;
}

// holds one CU per workgroup (all of its LDS) until `ticks` of the 100 MHz wall clock have passed
__global__ void __launch_bounds__(512) kh_occupy_kernel(long long ticks, int *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long t0 = wall_clock64();
    // all ones (a NaN as a double) in all of the 128 KiB: what a later kernel on this CU finds in its uninitialised LDS
    for (int i = threadIdx.x; i < 128 * 1024 / 16; i += blockDim.x) ((uint4 *)smem)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    __syncthreads();
    smem[threadIdx.x] = (char)threadIdx.x;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (smem[(threadIdx.x + 1) & 511] == 77 && ticks < 0) *sink = 1;
}

extern "C" int kh_debug_occupy(kh_engine *e, int32_t workgroups, double milliseconds, void *stream) {
    if (e == nullptr || workgroups < 1 || !(milliseconds >= 0.0) || milliseconds > 1000.0)
        return kh_fail(KH_ERR_INVALID, "bad argument");
    const size_t lds = 128 * 1024;
    const int rc = ensure_dynamic_lds(e, (const void *)kh_occupy_kernel, lds);
    if (rc != KH_OK) return rc;
    kh_occupy_kernel<<<workgroups, 512, lds, (hipStream_t)stream>>>((long long)(milliseconds * 1e5), (int *)e->d_abort + 0);
    KH_HIP(hipGetLastError());
    return KH_OK;
}

extern "C" int kh_debug_launched(int32_t which, char *buf, int32_t cap) {
    std::string out;
    if (which == 2) {  // forget what this process has launched so far (not what it has logged)
        for (KhKernelRecord &rec : kh_kernel_registry()) rec.launched = false;
        return 0;
    }
    for (const KhKernelRecord &rec : kh_kernel_registry())
        if (which != 0 || rec.launched) out += rec.name + "\n";
    if (buf != nullptr && cap > 0) {
        const size_t n = out.size() < (size_t)cap - 1 ? out.size() : (size_t)cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int)out.size() + 1;
}

extern "C" int kh_last_stats(kh_engine *e, double stats[4]) {
    if (e == nullptr || stats == nullptr) return kh_fail(KH_ERR_INVALID, "null argument");
    double d[4] = {0, 0, 0, 0};
    KH_HIP(hipMemcpy(d, e->d_stats, sizeof(d), hipMemcpyDeviceToHost));
    stats[0] = d[0];
    stats[1] = e->last_intervals;
    stats[2] = e->last_wgs;
    stats[3] = 0.0;
#ifdef KH_TIMING
    stats[0] = d[0];
    stats[1] = d[1];
    stats[2] = d[2];
    stats[3] = d[3];
    if (getenv("KH_TRACE")) {  // per-point clock stamps of one interval of workgroup 0 (kernels that record them)
        double tr[64];
        KH_HIP(hipMemcpy(tr, e->d_stats + 4, sizeof(tr), hipMemcpyDeviceToHost));
        fprintf(stderr, "KH_TRACE (cycles since the interval's start):");
        for (int i = 0; i < 64 && (i == 0 || tr[i] != 0.0); ++i) fprintf(stderr, " %d:%.0f", i, tr[i] - tr[0]);
        fprintf(stderr, "\n");
        unsigned int ab[2] = {0, 0};
        KH_HIP(hipMemcpy(ab, e->d_abort, sizeof(ab), hipMemcpyDeviceToHost));
        fprintf(stderr, "KH_TRACE polling rounds of workgroup 0 since the engine was created: %u\n", ab[1]);
        fprintf(stderr, "KH_TRACE raw [16..31]:");
        for (int i = 16; i < 32; ++i) fprintf(stderr, " %.0f", tr[i]);
        fprintf(stderr, "\n");
    }
#endif
    return KH_OK;
}

extern "C" int32_t kh_ell_rows_of(int32_t N) { return N >= 1 && N <= KH_ELL_NMAX ? kh_ell_rows(N) : 0; }

extern "C" int kh_ell_layout(int32_t N, int32_t n_ops, const kh_csr *ops_host, int32_t *E_out, int32_t *Ec_out, int32_t *off_out,
                             kh_cdouble *vals_out, int32_t E_cap) {
    if (ops_host == nullptr || E_out == nullptr || Ec_out == nullptr || n_ops < 1 || N < 1)
        return kh_fail(KH_ERR_INVALID, "bad argument");
    if (N > KH_ELL_NMAX) return kh_fail(KH_ERR_UNSUPPORTED, "N = %d: the padded row form serves N <= %d", N, KH_ELL_NMAX);
    std::vector<HostCsr> host(n_ops);
    std::vector<const HostCsr *> ptrs(n_ops, nullptr);
    for (int o = 0; o < n_ops; ++o) {
        if (ops_host[o].data == nullptr) continue;
        if (!canonical_csr(ops_host[o].indptr, ops_host[o].indices, (const cplx *)ops_host[o].data, ops_host[o].nnz, N, host[o]))
            return kh_fail(KH_ERR_INVALID, "operator %d: inconsistent CSR arrays", o);
        ptrs[o] = &host[o];
    }
    std::vector<int> off;
    std::vector<cplx> vals;
    int E = 0, Ec = 0;
    if (!build_ell_host(ptrs, N, off, vals, E, Ec))
        return kh_fail(KH_ERR_UNSUPPORTED, "rows wider than the kernels' register budget (%d entries for N = %d)", kh_ell_emax(N), N);
    *E_out = E;
    *Ec_out = Ec;
    if (off_out != nullptr || vals_out != nullptr) {
        if (E_cap < E) return kh_fail(KH_ERR_INVALID, "E_cap = %d < E = %d", E_cap, E);
        if (off_out != nullptr) memcpy(off_out, off.data(), sizeof(int) * off.size());
        if (vals_out != nullptr) memcpy(vals_out, vals.data(), sizeof(cplx) * vals.size());
    }
    return KH_OK;
}

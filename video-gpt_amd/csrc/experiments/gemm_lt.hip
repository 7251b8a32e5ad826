// The vendor GEMM library (hipBLASLt) behind vgpt_gemm_bf16 / vgpt_gemm_bf16_tr for PLAIN products: C = A W^T with nothing
// fused beyond what a library GEMM offers itself (beta = 1 for the residual, a bias vector).  The hand-written kernels of
// gemm_bf16.hip keep everything that IS fused (RoPE in the qkv epilogue, the gated activation, the training forward's
// gate_up store) and every shape the table below does not name; this file only forwards the two products of a decoder layer
// that are library GEMMs as they stand -- o_proj and down_proj, N = hidden -- because hipBLASLt's stream-K assembly kernel
// (four waves, a 96 x 128 accumulator per wave) is 14-17 % ahead of the 8-wave HIP kernel on exactly those (DESIGN.md 4b,
// profiles/r03_vendor_gemm_ab.json; it is level or behind on the wide ones).
//
// No link-time dependency: the library is looked up with dlopen when the first such product arrives -- the copy already in the
// process if there is one (PyTorch ships its own next to its HIP runtime, and two hipBLASLt builds in one process are one
// too many), else the ROCm installation's -- and only nine C entry points are bound.  If it is absent, disabled
// (VGPT_GEMM_VENDOR=0) or refuses a problem, the caller launches the hand-written kernel: same operands, same epilogue.
#include "common.h"

#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>   // types and enums; every function is called through a dlsym pointer
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <tuple>

namespace {

struct Api {
    void* so = nullptr;
    hipblasStatus_t (*Create)(hipblasLtHandle_t*) = nullptr;
    hipblasStatus_t (*LayoutCreate)(hipblasLtMatrixLayout_t*, hipDataType, uint64_t, uint64_t, int64_t) = nullptr;
    hipblasStatus_t (*DescCreate)(hipblasLtMatmulDesc_t*, hipblasComputeType_t, hipDataType) = nullptr;
    hipblasStatus_t (*DescSet)(hipblasLtMatmulDesc_t, hipblasLtMatmulDescAttributes_t, const void*, size_t) = nullptr;
    hipblasStatus_t (*PrefCreate)(hipblasLtMatmulPreference_t*) = nullptr;
    hipblasStatus_t (*PrefSet)(hipblasLtMatmulPreference_t, hipblasLtMatmulPreferenceAttributes_t, const void*, size_t) = nullptr;
    hipblasStatus_t (*PrefDestroy)(const hipblasLtMatmulPreference_t) = nullptr;   // optional: a missing symbol leaks a few bytes per shape
    hipblasStatus_t (*Heuristic)(hipblasLtHandle_t, hipblasLtMatmulDesc_t, hipblasLtMatrixLayout_t, hipblasLtMatrixLayout_t,
                                 hipblasLtMatrixLayout_t, hipblasLtMatrixLayout_t, hipblasLtMatmulPreference_t, int,
                                 hipblasLtMatmulHeuristicResult_t*, int*) = nullptr;
    hipblasStatus_t (*Matmul)(hipblasLtHandle_t, hipblasLtMatmulDesc_t, const void*, const void*, hipblasLtMatrixLayout_t,
                              const void*, hipblasLtMatrixLayout_t, const void*, const void*, hipblasLtMatrixLayout_t, void*,
                              hipblasLtMatrixLayout_t, const hipblasLtMatmulAlgo_t*, void*, size_t, hipStream_t) = nullptr;
    char origin[160] = "not loaded";
};

constexpr size_t WORKSPACE_BYTES = 128u << 20;   // stream-K partial sums and flags of the library's kernels

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr, d = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws = 0;
    bool ok = false;
};
using PlanKey = std::tuple<int, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int, int>;

struct State {
    std::mutex mu;
    int mode = -1;       // -1 undecided, 0 off, 1 table ("auto"), 2 every plain product it accepts ("all": probes and tests)
    bool tried = false;
    Api api;
    std::map<int, hipblasLtHandle_t> handles;                  // per device
    std::map<std::pair<int, hipStream_t>, void*> workspaces;   // per (device, stream): concurrent streams never share partials
    std::map<PlanKey, Plan> plans;
    long calls = 0;
};
State& st() {
    static State* s = new State;   // never destroyed: the library may be torn down after us at process exit
    return *s;
}

int decide_mode() {
    const char* e = getenv("VGPT_GEMM_VENDOR");
    if (!e || !*e || !strcmp(e, "auto") || !strcmp(e, "1")) return 1;
    if (!strcmp(e, "0") || !strcmp(e, "off")) return 0;
    if (!strcmp(e, "all")) return 2;
    return 1;
}

template <typename F>
bool bind(void* so, F& f, const char* name) {
    f = (F)dlsym(so, name);
    return f != nullptr;
}

bool load_api(Api& a) {
    const char* forced = getenv("VGPT_HIPBLASLT_PATH");
    const char* names[] = {"libhipblaslt.so", "libhipblaslt.so.1"};
    void* so = nullptr;
    if (forced && *forced) {
        so = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (so) snprintf(a.origin, sizeof a.origin, "VGPT_HIPBLASLT_PATH=%s", forced);
    }
    for (int i = 0; !so && i < 2; ++i) {   // the copy this process already holds
        so = dlopen(names[i], RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (so) snprintf(a.origin, sizeof a.origin, "%s (already in the process)", names[i]);
    }
    for (int i = 1; !so && i >= 0; --i) {  // else the installation's
        so = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (so) snprintf(a.origin, sizeof a.origin, "%s (loaded from the library path)", names[i]);
    }
    if (!so) {
        snprintf(a.origin, sizeof a.origin, "hipBLASLt not found");
        return false;
    }
    const bool ok = bind(so, a.Create, "hipblasLtCreate") && bind(so, a.LayoutCreate, "hipblasLtMatrixLayoutCreate") &&
                    bind(so, a.DescCreate, "hipblasLtMatmulDescCreate") && bind(so, a.DescSet, "hipblasLtMatmulDescSetAttribute") &&
                    bind(so, a.PrefCreate, "hipblasLtMatmulPreferenceCreate") &&
                    bind(so, a.PrefSet, "hipblasLtMatmulPreferenceSetAttribute") &&
                    bind(so, a.Heuristic, "hipblasLtMatmulAlgoGetHeuristic") && bind(so, a.Matmul, "hipblasLtMatmul");
    if (!ok) {
        snprintf(a.origin, sizeof a.origin, "hipBLASLt found but an entry point is missing");
        return false;
    }
    (void)bind(so, a.PrefDestroy, "hipblasLtMatmulPreferenceDestroy");
    a.so = so;
    return true;
}

// The products the library takes in "auto" mode (scripts/gemm_vendor_probe.py, profiles/r03_vendor_gemm_probe.jsonl; same box,
// hand-written kernel = 1.00):
//   C = A W^T (+ residual | + bias) onto 3072-wide outputs, K = 3072 / 8192 -- o_proj and down_proj of the 3.8 B decoder:
//       1.30-1.50 at 1448-1848 rows (per-clip pass), 1.04-1.18 at 4096 (sampler step), 1.13-1.20 at 7740-8192 (training)
//   dX = dY W with both widths <= 4096 (o_proj's input gradient): 1.09-1.12
// and does not take: 9216-wide outputs at 7.7 k rows (0.91-0.95), dW = dY^T X (0.73-0.92 on the wide ones), the input
// gradients of the wide layers (1.01-1.06: not worth a second code path's rounding), and of course nothing fused.
//   purpose 1, the training forward's [gate | up] = A W^T that has to be STORED anyway (16384-wide): the library's plain GEMM
//       + the activation kernel 591 + 59 us at 7740 rows against 722 for the fused kernel that also stores [gate | up]
bool table_says_vendor(int64_t M, int64_t N, int64_t K, bool a_tr, bool w_tr, int purpose) {
    if (purpose == 1) return !a_tr && !w_tr && M >= 2048 && N >= 8192 && K >= 1024;
    if (a_tr) return false;
    if (M < 1024 || N < 1024 || N > 4096 || K < 1024) return false;
    return !w_tr || K <= 4096;
}

bool capturing(hipStream_t s) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return cs != hipStreamCaptureStatusNone;
}

}  // namespace

// 1: the product was enqueued on the library's kernel; 0: not taken (the caller launches its own kernel).  Never an error:
// whatever the library cannot or may not do, gemm_bf16.hip can.
int vgpt_lt_try_gemm(const void* A, const void* W, void* C, const void* extra, int64_t M, int64_t N, int64_t K, int64_t lda,
                     int64_t ldw, int64_t ldc, int64_t ldr, int epilogue, int a_tr, int w_tr, hipStream_t stream, int purpose) {
    State& s = st();
    std::lock_guard<std::mutex> lock(s.mu);
    if (s.mode < 0) s.mode = decide_mode();
    if (s.mode == 0) return 0;
    if (s.mode == 1 && !table_says_vendor(M, N, K, a_tr, w_tr, purpose)) return 0;
    if (M < 16 || N < 16 || K < 16) return 0;
    // the library's kernels move 16 bytes per lane: rows it may not assume aligned stay on the hand-written kernel
    // (vgpt_gemm_bf16 itself only asks 8-byte alignment of C and of the residual)
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (epilogue == VGPT_EPI_NONE ? 0 : (uintptr_t)extra)) & 15) != 0 ||
        ((lda | ldw | ldc | (epilogue == VGPT_EPI_RESID ? ldr : 0)) & 7) != 0)
        return 0;
    if (!s.tried) {
        s.tried = true;
        if (!load_api(s.api)) s.mode = 0;
    }
    if (!s.api.so) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const bool cap = capturing(stream);

    auto hit = s.handles.find(dev);
    if (hit == s.handles.end()) {
        if (cap) return 0;
        hipblasLtHandle_t h = nullptr;
        if (s.api.Create(&h) != HIPBLAS_STATUS_SUCCESS || !h) return 0;
        hit = s.handles.emplace(dev, h).first;
    }
    auto wkey = std::make_pair(dev, stream);
    auto wit = s.workspaces.find(wkey);
    if (wit == s.workspaces.end()) {
        if (cap) return 0;    // an allocation cannot be recorded: the first product of a stream must come eagerly
        void* p = nullptr;
        if (hipMalloc(&p, WORKSPACE_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        wit = s.workspaces.emplace(wkey, p).first;
    }

    const PlanKey key{dev, M, N, K, lda, ldw, ldc, epilogue == VGPT_EPI_RESID ? ldr : 0, epilogue, a_tr, w_tr};
    auto pit = s.plans.find(key);
    if (pit == s.plans.end()) {
        Plan p;
        // row-major C (M, N) is the column-major (N, M) matrix D = op(W memory) op(A memory)
        const hipblasOperation_t opa = w_tr ? HIPBLAS_OP_N : HIPBLAS_OP_T, opb = a_tr ? HIPBLAS_OP_T : HIPBLAS_OP_N;
        bool ok = s.api.DescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS;
        ok = ok && s.api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof opa) == HIPBLAS_STATUS_SUCCESS;
        ok = ok && s.api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof opb) == HIPBLAS_STATUS_SUCCESS;
        if (ok && epilogue == VGPT_EPI_BIAS) {
            const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
            const hipDataType bt = HIP_R_16BF;
            ok = s.api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof epi) == HIPBLAS_STATUS_SUCCESS &&
                 s.api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof bt) == HIPBLAS_STATUS_SUCCESS &&
                 s.api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &extra, sizeof extra) == HIPBLAS_STATUS_SUCCESS;
        }
        ok = ok && s.api.LayoutCreate(&p.a, HIP_R_16BF, w_tr ? N : K, w_tr ? K : N, ldw) == HIPBLAS_STATUS_SUCCESS;
        ok = ok && s.api.LayoutCreate(&p.b, HIP_R_16BF, a_tr ? M : K, a_tr ? K : M, lda) == HIPBLAS_STATUS_SUCCESS;
        ok = ok && s.api.LayoutCreate(&p.c, HIP_R_16BF, N, M, epilogue == VGPT_EPI_RESID ? ldr : ldc) == HIPBLAS_STATUS_SUCCESS;
        ok = ok && s.api.LayoutCreate(&p.d, HIP_R_16BF, N, M, ldc) == HIPBLAS_STATUS_SUCCESS;
        hipblasLtMatmulPreference_t pref = nullptr;
        ok = ok && s.api.PrefCreate(&pref) == HIPBLAS_STATUS_SUCCESS;
        const uint64_t wsmax = WORKSPACE_BYTES;
        ok = ok && s.api.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsmax, sizeof wsmax) == HIPBLAS_STATUS_SUCCESS;
        if (ok) {
            // the library's heuristic ranks its kernels without running them; VGPT_GEMM_VENDOR_TUNE=N (N <= 16) runs its first
            // N candidates on the caller's operands (output into a scratch buffer) when a shape is first seen OUTSIDE a capture
            // and keeps the fastest (=2 and up also prints the times).  Off by default: the choice would depend on the box.
            constexpr int MAXC = 16;
            static int tune = -1;
            if (tune < 0) { const char* e = getenv("VGPT_GEMM_VENDOR_TUNE"); tune = e ? atoi(e) : 0; if (tune > MAXC) tune = MAXC; }
            const int want = (tune > 1 && !cap) ? tune : 1;
            hipblasLtMatmulHeuristicResult_t res[MAXC];
            int got = 0;
            ok = s.api.Heuristic(hit->second, p.desc, p.a, p.b, p.c, p.d, pref, want, res, &got) == HIPBLAS_STATUS_SUCCESS &&
                 got >= 1;
            int best = -1;
            for (int i = 0; ok && i < got && best < 0; ++i)
                if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= WORKSPACE_BYTES) best = i;
            ok = ok && best >= 0;
            if (ok && want > 1 && got > 1) {
                void* scratch = nullptr;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (hipMalloc(&scratch, (size_t)M * (size_t)ldc * 2) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
                    hipEventCreate(&e1) == hipSuccess) {
                    const float one = 1.0f, zero = 0.0f;
                    const bool resid = epilogue == VGPT_EPI_RESID;
                    float best_ms = 1e30f;
                    for (int i = 0; i < got; ++i) {
                        if (res[i].state != HIPBLAS_STATUS_SUCCESS || res[i].workspaceSize > WORKSPACE_BYTES) continue;
                        bool fine = true;
                        for (int rep = 0; rep < 5 && fine; ++rep) {   // 2 untimed, 3 timed
                            if (rep == 2) (void)hipEventRecord(e0, stream);
                            fine = s.api.Matmul(hit->second, p.desc, &one, W, p.a, A, p.b, resid ? &one : &zero,
                                                resid ? extra : scratch, p.c, scratch, p.d, &res[i].algo, wit->second,
                                                WORKSPACE_BYTES, stream) == HIPBLAS_STATUS_SUCCESS;
                        }
                        (void)hipEventRecord(e1, stream);
                        (void)hipEventSynchronize(e1);
                        float ms = 0.f;
                        if (!fine || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
                        if (tune >= 3)
                            fprintf(stderr, "[vgpt gemm_lt] %ld x %ld x %ld epi %d tr %d%d: candidate %d %.1f us\n", (long)M, (long)N,
                                    (long)K, epilogue, a_tr, w_tr, i, ms / 3 * 1e3);
                        if (ms < best_ms) { best_ms = ms; best = i; }
                    }
                    if (tune >= 2)
                        fprintf(stderr, "[vgpt gemm_lt] %ld x %ld x %ld epi %d tr %d%d: candidate %d of %d kept (%.1f us)\n", (long)M,
                                (long)N, (long)K, epilogue, a_tr, w_tr, best, got, best_ms / 3 * 1e3);
                }
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
                if (scratch) (void)hipFree(scratch);
                (void)hipGetLastError();
            }
            if (ok) {
                p.algo = res[best].algo;
                p.ws = res[best].workspaceSize;
            }
        }
        if (pref && s.api.PrefDestroy) (void)s.api.PrefDestroy(pref);
        p.ok = ok;    // a refused problem is remembered too: it goes to the hand-written kernel from now on
        pit = s.plans.emplace(key, p).first;
    }
    Plan& p = pit->second;
    if (!p.ok) return 0;
    if (epilogue == VGPT_EPI_BIAS &&
        s.api.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &extra, sizeof extra) != HIPBLAS_STATUS_SUCCESS)
        return 0;
    const float one = 1.0f, zero = 0.0f;
    const bool resid = epilogue == VGPT_EPI_RESID;
    const hipblasStatus_t rc = s.api.Matmul(hit->second, p.desc, &one, W, p.a, A, p.b, resid ? &one : &zero, resid ? extra : C,
                                            p.c, C, p.d, &p.algo, wit->second, WORKSPACE_BYTES, stream);
    if (rc != HIPBLAS_STATUS_SUCCESS) {
        p.ok = false;
        return 0;
    }
    ++s.calls;
    return 1;
}

/* ---- what the library is doing, for bench.py / tests (include/vgpt.h) ---- */
VGPT_EXPORT int vgpt_gemm_vendor_applies(int64_t M, int64_t N, int64_t K, int a_transposed, int w_transposed) {
    State& s = st();
    std::lock_guard<std::mutex> lock(s.mu);
    if (s.mode < 0) s.mode = decide_mode();
    if (s.mode == 0) return 0;
    if (s.mode == 1) return table_says_vendor(M, N, K, a_transposed, w_transposed, 0) ? 1 : 0;
    return M >= 16 && N >= 16 && K >= 16;
}

VGPT_EXPORT int vgpt_gemm_vendor_ready(void* stream) {
    State& s = st();
    std::lock_guard<std::mutex> lock(s.mu);
    int dev = 0;
    if (!s.api.so || hipGetDevice(&dev) != hipSuccess) return 0;
    return s.handles.count(dev) && s.workspaces.count(std::make_pair(dev, (hipStream_t)stream)) ? 1 : 0;
}

VGPT_EXPORT int64_t vgpt_gemm_vendor_calls(void) {
    State& s = st();
    std::lock_guard<std::mutex> lock(s.mu);
    return s.calls;
}

VGPT_EXPORT const char* vgpt_gemm_vendor_origin(void) {
    State& s = st();
    std::lock_guard<std::mutex> lock(s.mu);
    if (s.mode < 0) s.mode = decide_mode();
    if (s.mode == 0 && !s.tried) return "disabled (VGPT_GEMM_VENDOR=0)";
    return s.api.origin;
}

/* mode: 0 off, 1 the table, 2 every plain product (probes / tests); returns the previous mode.  Takes effect on the next call;
 * plans already made stay cached. */
VGPT_EXPORT int vgpt_gemm_vendor_set_mode(int mode) {
    State& s = st();
    std::lock_guard<std::mutex> lock(s.mu);
    if (s.mode < 0) s.mode = decide_mode();
    const int prev = s.mode;
    if (mode >= 0 && mode <= 2) {
        s.mode = mode;
        if (mode > 0 && s.tried && !s.api.so) s.mode = 0;   // looked for and not found: stays off
    }
    return prev;
}

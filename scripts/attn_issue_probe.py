"""Developer probe: what one SIMD of gfx950 can ISSUE per attention tile.  The forward kernel's tile loop (32 query rows x 64
keys per wave, head dim 96) holds 24 v_mfma_f32_32x32x16_bf16, 36 LDS fragment reads and the softmax's vector mix counted
from the code object (33 v_exp_f32, 24 v_pk_mul_f32, 19 v_pk_add_f32, 16 v_pk_fma_f32, 16 v_max3_f32, 16
v_cvt_pk_bf16_f32, ~30 one-cycle-class integer / move instructions).  The matrix pipe is busy 40 % of the kernel
(profiles/r04_pmc_mfma.json): this probe measures, with s_memtime around straight-line asm loops and NO memory traffic, how
many cycles each ingredient costs alone, what the sum costs in program order with the real register dependencies
(MFMA -> softmax -> MFMA), what a perfect interleave inside one wave costs, and what a SECOND wave on the SIMD recovers.

  python scripts/attn_issue_probe.py --build     (here: writes csrc/experiments/issue_probe_gen.hip, compiles libvgpt_x_issue.so)
  python scripts/attn_issue_probe.py             (GPU box: runs every variant at 1 and 2 waves per SIMD, JSON lines)
"""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "video-gpt_amd", "csrc")
SO = os.path.join(ROOT, "video-gpt_amd", "libvgpt_x_issue.so")

# register map (physical, named in the asm): v1 LDS address; v[2:5] / v[6:9] constant MFMA operands; v[10:13] scalars of the
# softmax (scale pair, reference pair); v[20:23] running maxima; v[28:35] row-sum chains; S^T accumulators v[40:71] (two
# 32x32 blocks); P (bf16 pairs) v[130:145]; O^T accumulators v[80:127] (three 32x32 blocks); K fragments v[150:197] (12 x 4),
# V fragments in the same registers (24 x 2)
S0, P0, O0, KF, VF = 40, 130, 80, 150, 150      # (the V fragments reuse the K fragments' registers: 248 registers, two waves per SIMD)


def mfma_qk(i):      # 2 accumulators x 6 k-steps, alternating; operands = K fragment i (A) and the resident Q fragment (B)
    acc = S0 + 16 * (i & 1)
    return f"v_mfma_f32_32x32x16_bf16 v[{acc}:{acc + 15}], v[{KF + 4 * i}:{KF + 4 * i + 3}], v[6:9], v[{acc}:{acc + 15}]"


def mfma_pv(i):      # 3 accumulators x 4 key steps; A = V fragment pair (two 8-byte transposed reads), B = P of that key step
    acc = O0 + 16 * (i % 3)
    ks = i // 3
    return f"v_mfma_f32_32x32x16_bf16 v[{acc}:{acc + 15}], v[{VF + 4 * (i % 12)}:{VF + 4 * (i % 12) + 3}], v[{P0 + 4 * ks}:{P0 + 4 * ks + 3}], v[{acc}:{acc + 15}]"


def valu_mix():
    """The softmax of one tile in dependency order: maxima of the 32 scores, exp2(scale * s - m), row sums, rounding to bf16,
    rescale of the 48 output accumulators."""
    mx = [f"v_max3_f32 v{20 + (i & 3)}, v{S0 + 2 * i}, v{S0 + 2 * i + 1}, v{20 + (i & 3)}" for i in range(16)]
    mx += [f"v_max_f32 v20, v20, v21", "v_max_f32 v22, v22, v23", "v_max_f32 v20, v20, v22",
           "v_mov_b32 v14, v20", "s_nop 1", "v_permlane32_swap_b32 v14, v20", "v_max_f32 v20, v20, v14", "v_sub_f32 v15, v12, v20", "v_exp_f32 v15, v15",
           "v_mov_b32 v16, v15", "v_mov_b32 v17, v15"]
    fma = [f"v_pk_fma_f32 v[{S0 + 2 * i}:{S0 + 2 * i + 1}], v[{S0 + 2 * i}:{S0 + 2 * i + 1}], v[10:11], v[12:13]" for i in range(16)]
    ex = [f"v_exp_f32 v{S0 + i}, v{S0 + i}" for i in range(32)]
    sm = [f"v_pk_add_f32 v[{28 + 2 * (i & 3)}:{29 + 2 * (i & 3)}], v[{28 + 2 * (i & 3)}:{29 + 2 * (i & 3)}], v[{S0 + 2 * i}:{S0 + 2 * i + 1}]" for i in range(16)]
    sm += ["v_pk_add_f32 v[28:29], v[28:29], v[30:31]", "v_pk_add_f32 v[32:33], v[32:33], v[34:35]", "v_pk_add_f32 v[28:29], v[28:29], v[32:33]"]
    cv = [f"v_cvt_pk_bf16_f32 v{P0 + i}, v{S0 + 2 * i}, v{S0 + 2 * i + 1}" for i in range(16)]
    rs = [f"v_pk_mul_f32 v[{O0 + 2 * i}:{O0 + 2 * i + 1}], v[{O0 + 2 * i}:{O0 + 2 * i + 1}], v[16:17]" for i in range(24)]
    misc = [f"v_add_u32 v{18 + (i & 1)}, v{18 + (i & 1)}, v1" for i in range(14)]
    return {"max": mx, "fma": fma, "exp": ex, "sum": sm, "cvt": cv, "rescale": rs, "misc": misc}


def lds_reads():
    k = [f"ds_read_b128 v[{KF + 4 * i}:{KF + 4 * i + 3}], v1 offset:{i * 1024}" for i in range(12)]
    v = [f"ds_read_b64_tr_b16 v[{VF + 2 * i}:{VF + 2 * i + 1}], v1 offset:{16384 + i * 512}" for i in range(24)]
    return k, v


def interleave(primary, fillers):
    """primary instructions in order, the fillers spread evenly between them"""
    out, n, m = [], len(primary), len(fillers)
    j = 0
    for i, p in enumerate(primary):
        out.append(p)
        upto = (i + 1) * m // n
        out += fillers[j:upto]
        j = upto
    return out


def variants():
    vm = valu_mix()
    softmax = vm["max"] + vm["fma"] + vm["exp"] + vm["sum"] + vm["cvt"] + vm["rescale"] + vm["misc"]
    k, v = lds_reads()
    qk = [mfma_qk(i) for i in range(12)]
    pv = [mfma_pv(i) for i in range(12)]
    wait = ["s_waitcnt lgkmcnt(0)"]
    V = {}
    V["mfma24"] = qk + pv
    V["softmax_all"] = softmax
    for name in vm:
        V["valu_" + name] = vm[name]
    V["lds36"] = k + v + wait
    # the kernel's order: K fragments, QK^T, V requests, softmax, P.V -- registers carry the real dependencies; the s_nop stand for
    # the wait states the compiler must put between a matrix result and its vector reader (8-pass MFMA) and a vector result and
    # its matrix reader
    gap, gap2 = ["s_nop 7", "s_nop 3"], ["s_nop 1"]
    V["program_order"] = k + wait + qk + v + gap + softmax + wait + gap2 + pv
    V["program_order_no_lds"] = qk + gap + softmax + gap2 + pv
    # a naive interleave inside the wave (same registers: every dependency stall it implies is in the number) ...
    V["interleaved_dep"] = interleave(qk + pv, softmax)
    # ... and the ideal one: the MFMAs accumulate in registers of their own (v[200:247]; AGPRs would halve the
    # vector registers hipcc grants at two waves per SIMD) from constant operands, nothing they touch is touched by the vector
    # stream -- what a perfectly software-pipelined wave would pay for issue alone
    free = [f"v_mfma_f32_32x32x16_bf16 v[{200 + 16 * (i % 3)}:{215 + 16 * (i % 3)}], v[2:5], v[6:9], v[{200 + 16 * (i % 3)}:{215 + 16 * (i % 3)}]" for i in range(24)]
    V["mfma24_free"] = free
    V["interleaved_free"] = interleave(free, softmax)
    V["interleaved_free_lds"] = interleave(free, softmax + k + v) + wait
    V["phases_free"] = free[:12] + softmax + free[12:]
    V["phases_free_lds"] = k + wait + free[:12] + v + softmax + wait + free[12:]
    # co-issue ladders: one independent MFMA followed by N independent vector instructions of one kind, 24 groups per iteration
    def op(kind, j):
        r = 40 + 2 * (j % 40)
        return {"add": f"v_add_u32 v{r}, v{r}, v1", "exp": f"v_exp_f32 v{r}, v{r}", "pkfma": f"v_pk_fma_f32 v[{r}:{r + 1}], v[{r}:{r + 1}], v[10:11], v[12:13]",
                "pkmul": f"v_pk_mul_f32 v[{r}:{r + 1}], v[{r}:{r + 1}], v[16:17]", "cvt": f"v_cvt_pk_bf16_f32 v{r}, v{r}, v{r + 1}",
                "max3": f"v_max3_f32 v{r}, v{r}, v{r + 1}, v{r}", "fma": f"v_fma_f32 v{r}, v{r}, v10, v12"}[kind]
    for kind in ("add", "fma", "exp", "pkfma", "pkmul", "cvt", "max3"):
        for n in (2, 4, 6, 8, 12):
            body, j = [], 0
            for g in range(24):
                body.append(free[g])
                for _ in range(n):
                    body.append(op(kind, j)); j += 1
            V[f"coissue_{kind}_{n}"] = body
    small = [f"v_mfma_f32_16x16x32_bf16 v[{200 + 4 * (i % 12)}:{203 + 4 * (i % 12)}], v[2:5], v[6:9], v[{200 + 4 * (i % 12)}:{203 + 4 * (i % 12)}]" for i in range(48)]
    V["mfma16_48"] = small
    for n in (1, 2, 3, 4, 6):
        body, j = [], 0
        for g in range(48):
            body.append(small[g])
            for _ in range(n):
                body.append(op("add", j)); j += 1
        V[f"coissue16_add_{n}"] = body
    return V


def split_variants():
    """waves 0..3 of an eight-wave workgroup run the first body, waves 4..7 (the second wave of each SIMD) the second"""
    V = variants()
    return {"split_mfma_softmax": (V["mfma24_free"], V["softmax_all"]), "split_mfma_exp": (V["mfma24_free"], V["valu_exp"] * 2),
            "split_mfma_add": (V["mfma24_free"], [f"v_add_u32 v{40 + (i % 80)}, v{40 + (i % 80)}, v1" for i in range(192)]),
            "split_softmax_softmax": (V["softmax_all"], V["softmax_all"])}


HEADER = r"""// GENERATED by scripts/attn_issue_probe.py --build -- issue-rate probe of the attention tile's instruction mix (timing only,
// the arithmetic is meaningless).  Not part of libvgpt_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define CLOBBERS %s
"""

KERNEL = r"""
extern "C" __global__ __launch_bounds__(256, 2) void probe_%(name)s(unsigned long long* out, int iters) {
    extern __shared__ char smem[];
    const uint32_t addr = (uint32_t)(uintptr_t)smem + (threadIdx.x & 63) * 16;
    __syncthreads();
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    asm volatile(
        "v_mov_b32 v1, %%[addr]\n"
        "v_mov_b32 v2, 0x3f803f80\n v_mov_b32 v3, v2\n v_mov_b32 v4, v2\n v_mov_b32 v5, v2\n"
        "v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n"
        "v_mov_b32 v10, 0x3f000000\n v_mov_b32 v11, v10\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n"
        "s_mov_b32 s20, %%[iters]\n"
        "s_nop 4\n"
        "1:\n"
%(body)s
        "s_sub_u32 s20, s20, 1\n"
        "s_cmp_lg_u32 s20, 0\n"
        "s_cbranch_scc1 1b\n"
        "s_nop 7\n s_nop 7\n"
        :: [addr] "v"(addr), [iters] "s"(iters) : CLOBBERS);
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = w1 - w0;
    }
}
"""

KERNEL2 = r"""
extern "C" __global__ __launch_bounds__(512, 1) void probe_%(name)s(unsigned long long* out, int iters) {
    extern __shared__ char smem[];
    const uint32_t addr = (uint32_t)(uintptr_t)smem + (threadIdx.x & 63) * 16;
    __syncthreads();
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    if (threadIdx.x < 256) {
        asm volatile(
            "v_mov_b32 v1, %%[addr]\n v_mov_b32 v2, 0x3f803f80\n v_mov_b32 v3, v2\n v_mov_b32 v4, v2\n v_mov_b32 v5, v2\n"
            "v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n"
            "v_mov_b32 v10, 0x3f000000\n v_mov_b32 v11, v10\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n"
            "s_mov_b32 s20, %%[iters]\n s_nop 4\n 1:\n"
%(body0)s
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n s_nop 7\n s_nop 7\n"
            :: [addr] "v"(addr), [iters] "s"(iters) : CLOBBERS);
    } else {
        asm volatile(
            "v_mov_b32 v1, %%[addr]\n v_mov_b32 v2, 0x3f803f80\n v_mov_b32 v3, v2\n v_mov_b32 v4, v2\n v_mov_b32 v5, v2\n"
            "v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n"
            "v_mov_b32 v10, 0x3f000000\n v_mov_b32 v11, v10\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n"
            "s_mov_b32 s20, %%[iters]\n s_nop 4\n 1:\n"
%(body1)s
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n s_nop 7\n s_nop 7\n"
            :: [addr] "v"(addr), [iters] "s"(iters) : CLOBBERS);
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = w1 - w0;
    }
}
"""

HOST = r"""
typedef void (*kern_t)(unsigned long long*, int);
static kern_t g_kernels[] = {%(ptrs)s};
extern "C" __attribute__((visibility("default"))) int vgptx_issue_probe(int id, int lds, int grid, int iters, unsigned long long* host_out) {
    const int threads = id >= %(nsingle)d ? 512 : 256;
    unsigned long long* d = nullptr;
    const size_t n = (size_t)grid * (threads / 64) * 2;
    if (hipMalloc(&d, n * 8) != hipSuccess) return -1;
    hipFuncSetAttribute((const void*)g_kernels[id], hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(g_kernels[id], dim3(grid), dim3(threads), lds, 0, d, iters);
    if (hipDeviceSynchronize() != hipSuccess) { hipFree(d); return -2; }
    hipMemcpy(host_out, d, n * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    return 0;
}
"""


def build():
    V = variants()
    clob = ", ".join(f'"v{i}"' for i in range(1, 250)) + ', "s20", "scc", "vcc", "memory"'
    src = HEADER % clob
    for name, body in V.items():
        src += KERNEL % {"name": name, "body": "\n".join(f'        "{line}\\n"' for line in body)}
    S2 = split_variants()
    fmt = lambda body: "\n".join(f'            "{line}\\n"' for line in body)
    for name, (b0, b1) in S2.items():
        src += KERNEL2 % {"name": name, "body0": fmt(b0), "body1": fmt(b1)}
    src += HOST % {"ptrs": ", ".join("probe_" + n for n in list(V) + list(S2)), "nsingle": len(V)}
    path = os.path.join(CSRC, "experiments", "issue_probe_gen.hip")
    open(path, "w").write(src)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", SO, path], check=True)
    print("built", SO, "variants:", ", ".join(V))


def run():
    V = list(variants())
    lib = ctypes.CDLL(SO)
    lib.vgptx_issue_probe.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p]
    iters = 2000
    for wps in (1, 2):
        # one workgroup of four waves per CU (LDS 100 KiB each) or two (64 KiB each: the forward kernel's occupancy)
        lds, grid, threads = (100 * 1024, 256, 256) if wps == 1 else (64 * 1024, 512, 256)
        for i, name in enumerate(V):
            n = grid * (threads // 64) * 2
            buf = (ctypes.c_ulonglong * n)()
            rc = lib.vgptx_issue_probe(i, lds, grid, iters, buf)
            if rc:
                print(json.dumps({"variant": name, "error": rc})); continue
            cyc = sorted(buf[0:n:2]); wall = sorted(buf[1:n:2])
            c = cyc[len(cyc) // 2] / iters
            ghz = cyc[len(cyc) // 2] / (wall[len(wall) // 2] * 10.0)     # wall clock = 100 MHz
            print(json.dumps({"variant": name, "waves_per_simd": wps, "cycles_per_tile_per_wave": round(c, 1),
                              "cycles_per_tile_per_simd": round(c / wps, 1), "clock_ghz": round(ghz, 3)}), flush=True)


def run_split():
    V, S2 = list(variants()), list(split_variants())
    lib = ctypes.CDLL(SO)
    lib.vgptx_issue_probe.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p]
    iters = 2000
    for i, name in enumerate(S2):
        n = 256 * 8 * 2
        buf = (ctypes.c_ulonglong * n)()
        rc = lib.vgptx_issue_probe(len(V) + i, 100 * 1024, 256, iters, buf)
        if rc:
            print(json.dumps({"variant": name, "error": rc})); continue
        first = sorted(buf[2 * (8 * b + w)] for b in range(256) for w in range(4))
        second = sorted(buf[2 * (8 * b + w)] for b in range(256) for w in range(4, 8))
        print(json.dumps({"variant": name, "waves_0_3_cycles_per_iter": round(first[len(first) // 2] / iters, 1),
                          "waves_4_7_cycles_per_iter": round(second[len(second) // 2] / iters, 1)}), flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        run()
        run_split()

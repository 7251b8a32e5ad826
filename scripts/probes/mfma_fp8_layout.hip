// Probe (GPU): operand lane map and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands, checked with
// exact small-integer data on an MI355X.  Findings (the layout csrc/attn_fp8.hip builds on):
//   * exp1: filling lane l (r = l & 31, h = l >> 5) of BOTH operands with the 32 k-values 32 h + j of row / column r gives
//     the exact product -- byte j of lane half h in A always meets byte j of lane half h in B;
//   * exp1b: with non-unit scales that simple reading is WRONG; what matches all 1024 outputs is: bytes 0..15 of a lane
//     are k = 16 h + j, bytes 16..31 are k = 32 + 16 h + j, and the E8M0 scale passed by lane (r, 0) covers k = 0..31 of
//     row r (bytes 0..15 of both lane halves), the one passed by lane (r, 1) covers k = 32..63;
//   * exp2 / exp3: one lane's scale moves exactly one row (A) or column (B) by half of its k range.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_fp8_layout mfma_fp8_layout.hip && ./mfma_fp8_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const unsigned char* A, const unsigned char* B, const unsigned char* sa, const unsigned char* sb, float* C) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v8i a, b;
    for (int w = 0; w < 8; ++w) {
        unsigned ua = 0, ub = 0;
        for (int t = 0; t < 4; ++t) {
            const int kk = 32 * h + 4 * w + t;
            ua |= (unsigned)A[r * 64 + kk] << (8 * t);
            ub |= (unsigned)B[kk * 32 + r] << (8 * t);
        }
        a[w] = (int)ua; b[w] = (int)ub;
    }
    v16f c = {};
    const int s_a = sa[r * 2 + h], s_b = sb[r * 2 + h];   // scale of (row r, k-block h) / (col r, k-block h)
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, s_a, 0, s_b);
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];   // row, col = lane & 31
}

static unsigned char e4m3_of_int(int v) {   // exact for |v| <= 8
    if (v == 0) return 0;
    unsigned char s = v < 0 ? 0x80 : 0; int m = abs(v);
    int e = 0; while ((1 << (e + 1)) <= m) ++e;            // m = 2^e * (1 + f)
    const int frac = (int)lround(((double)m / (1 << e) - 1.0) * 8);
    return s | (unsigned char)(((e + 7) << 3) | frac);
}

static int run(const std::vector<unsigned char>& A, const std::vector<unsigned char>& B, const std::vector<unsigned char>& sa,
               const std::vector<unsigned char>& sb, std::vector<float>& C) {
    unsigned char *dA, *dB, *dsa, *dsb; float* dC;
    (void)hipMalloc(&dA, A.size()); (void)hipMalloc(&dB, B.size()); (void)hipMalloc(&dsa, 64); (void)hipMalloc(&dsb, 64); (void)hipMalloc(&dC, 32 * 32 * 4);
    (void)hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
    C.resize(32 * 32);
    (void)hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dsa); (void)hipFree(dsb); (void)hipFree(dC);
    return 0;
}

int main() {
    std::vector<unsigned char> A(32 * 64), B(64 * 32), sa(64, 127), sb(64, 127);
    std::vector<int> Ai(32 * 64), Bi(64 * 32);
    std::vector<float> C;
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) { Ai[i] = rand() % 9 - 4; A[i] = e4m3_of_int(Ai[i]); Bi[i] = rand() % 9 - 4; B[i] = e4m3_of_int(Bi[i]); }
    // experiment 1: unit scales
    run(A, B, sa, sb, C);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0;
            for (int kk = 0; kk < 64; ++kk) ref += (double)Ai[i * 64 + kk] * Bi[kk * 32 + j];
            if (fabs(ref - C[i * 32 + j]) > 1e-3) { if (bad < 3) printf("exp1 mismatch [%d][%d]: got %g want %g\n", i, j, C[i * 32 + j], ref); ++bad; }
        }
    printf("exp1 (unit scales, k = 32h + j on both operands): %d mismatches\n", bad);
    for (int mode = 1; mode <= 3; ++mode) {
        std::vector<unsigned char> s1(64, 127), s2(64, 127);
        for (int i = 0; i < 64; ++i) { if (mode & 1) s1[i] = 127 + (i % 5) - 2; if (mode & 2) s2[i] = 127 + (i % 3) - 1; }
        run(A, B, s1, s2, C);
        bad = 0;
        int alt_ok = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0;
                for (int kk = 0; kk < 64; ++kk)
                    ref += (double)Ai[i * 64 + kk] * Bi[kk * 32 + j] * ldexp(1.0, s1[i * 2 + kk / 32] - 127) * ldexp(1.0, s2[j * 2 + kk / 32] - 127);
                // alternative model: byte j of lane half h is k = 16 h + j (j < 16) or 32 + 16 h + j - 16, scale blocks = k / 32
                double alt = 0;
                for (int kk = 0; kk < 64; ++kk) {
                    const int jj = kk & 31;
                    alt += (double)Ai[i * 64 + kk] * Bi[kk * 32 + j] * ldexp(1.0, s1[i * 2 + (jj >= 16)] - 127) * ldexp(1.0, s2[j * 2 + (jj >= 16)] - 127);
                }
                if (fabs(alt - C[i * 32 + j]) <= 1e-3) ++alt_ok;
                if (fabs(ref - C[i * 32 + j]) > 1e-3) { if (bad < 0) printf("exp1b mode %d mismatch [%d][%d]: got %g want %g\n", mode, i, j, C[i * 32 + j], ref); ++bad; }
            }
        printf("exp1b mode %d (1 = A scales vary, 2 = B scales vary): %d mismatches; alternative model matches %d of 1024\n", mode, bad, alt_ok);
    }
    // experiment 2: all ones, one lane's A scale doubled
    std::vector<unsigned char> ones(32 * 64, 0x38);
    for (int lane : {0, 5, 32, 37}) {
        std::vector<unsigned char> s2(64, 127);
        s2[(lane & 31) * 2 + (lane >> 5)] = 128;
        run(ones, ones, s2, sb, C);
        printf("exp2 lane %d scale 2x: ", lane);
        int shown = 0;
        for (int i = 0; i < 32 && shown < 6; ++i) for (int j = 0; j < 32 && shown < 6; ++j) if (C[i * 32 + j] != 64.f) { printf("C[%d][%d]=%g ", i, j, C[i * 32 + j]); ++shown; }
        int cnt = 0; for (float v : C) cnt += v != 64.f;
        printf(" (%d entries differ from 64)\n", cnt);
    }
    // experiment 3: one lane's B scale doubled
    for (int lane : {0, 37}) {
        std::vector<unsigned char> s2(64, 127);
        s2[(lane & 31) * 2 + (lane >> 5)] = 128;
        run(ones, ones, sa, s2, C);
        printf("exp3 lane %d B-scale 2x: ", lane);
        int shown = 0;
        for (int i = 0; i < 32 && shown < 4; ++i) for (int j = 0; j < 32 && shown < 4; ++j) if (C[i * 32 + j] != 64.f) { printf("C[%d][%d]=%g ", i, j, C[i * 32 + j]); ++shown; }
        int cnt = 0; for (float v : C) cnt += v != 64.f;
        printf(" (%d entries differ)\n", cnt);
    }
    return 0;
}

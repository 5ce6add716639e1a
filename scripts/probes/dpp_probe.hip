// dpp_probe.hip -- micro-benchmarks behind the round-4 decomposition of the forward kernel (k_fwd3): is the DP-ALU DPP form
//   v_fmac_f64_dpp vdst, src0 row_newbcast:k, src1      (vdst += src0[lane k of my row of 16] * src1)
// a full-rate fp64 FMA whose vector operand comes from ANOTHER LANE (no LDS operand stream, no butterfly)?  And what do the
// cross-row exchanges (v_permlane16_swap / v_permlane32_swap) cost?     build: hipcc --offload-arch=gfx950 -O3 dpp_probe.hip -o dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

#define FM(acc, x, m, k) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m))

__global__ void k_semantics(double *o, const double *a) {
    const double x = a[threadIdx.x], m = a[64 + threadIdx.x];
    double acc = 0.0;
    asm volatile("s_nop 4");
    FM(acc, x, m, 3);
    o[threadIdx.x] = acc;                   // expect x[(lane & ~15) + 3] * m[lane]
    double acc2 = 1.0;
    if (threadIdx.x & 1) { asm volatile("s_nop 4"); FM(acc2, x, m, 2); }        // even lanes inactive: source lane 2 is inactive in every row
    o[64 + threadIdx.x] = acc2;
    // sum over the four rows of 16 with the two swaps
    int lo = __double2loint(x), hi = __double2hiint(x);
    auto s16l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false); auto s16h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double x16 = x + __hiloint2double(s16h[0] == hi ? s16h[1] : s16h[0], s16l[0] == lo ? s16l[1] : s16l[0]);     // (debug form; the kernel uses the selection-free form below)
    o[128 + threadIdx.x] = x16;
    // selection-free: a = v, b = v;  swap16(a, b) -> a = [r0 r0 r2 r2], b = [r1 r1 r3 r3];  a + b = pair sums in both rows
    auto pl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false); auto ph = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double p = __hiloint2double(ph[0], pl[0]) + __hiloint2double(ph[1], pl[1]);
    int plo = __double2loint(p), phi = __double2hiint(p);
    auto ql = __builtin_amdgcn_permlane32_swap(plo, plo, false, false); auto qh = __builtin_amdgcn_permlane32_swap(phi, phi, false, false);
    o[192 + threadIdx.x] = __hiloint2double(qh[0], ql[0]) + __hiloint2double(qh[1], ql[1]);      // expect x[l%16] + x[l%16+16] + x[l%16+32] + x[l%16+48] in every lane
}

// throughput: NACC independent accumulators, REP x 16 FMAs each pass
template <int MODE, int NACC>
__global__ void __launch_bounds__(256) k_rate(double *o, const double *a, int rep, long long *cyc) {
    const double x = a[threadIdx.x & 63], m0 = a[64 + (threadIdx.x & 63)];
    double acc[NACC];
    double mm[8];
#pragma unroll
    for (int i = 0; i < 8; i++) mm[i] = m0 + i;
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = i;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rep; r++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if constexpr (MODE == 0) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(x), "v"(mm[u % 8])); }
            else if constexpr (MODE == 1) {
                switch (u) {
#define C(k) case k: FM(acc[k % NACC], x, mm[k % 8], k); break;
                    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
#undef C
                }
            } else if constexpr (MODE == 2) {       // mov_dpp + fma (what a 64-bit DPP move + plain FMA would cost)
                double t;
                asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(x));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(t), "v"(mm[u % 8]));
            } else if constexpr (MODE == 3) {       // cross-row reduction of a double over the 4 rows: 2 swaps x 2 halves + 2 adds
                int lo = __double2loint(acc[u % NACC]), hi = __double2hiint(acc[u % NACC]);
                auto pl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false); auto ph = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
                const double p = __hiloint2double(ph[0], pl[0]) + __hiloint2double(ph[1], pl[1]);
                int plo = __double2loint(p), phi = __double2hiint(p);
                auto ql = __builtin_amdgcn_permlane32_swap(plo, plo, false, false); auto qh = __builtin_amdgcn_permlane32_swap(phi, phi, false, false);
                acc[u % NACC] = (__hiloint2double(qh[0], ql[0]) + __hiloint2double(qh[1], ql[1])) * 0.25;
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int NACC>
static void run(const char *name, double *o, double *a, long long *cyc, int threads, int blocks) {
    const int rep = 2000;
    k_rate<MODE, NACC><<<blocks, threads>>>(o, a, rep, cyc); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k_rate<MODE, NACC><<<blocks, threads>>>(o, a, rep, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double ops = 16.0 * rep;
    printf("%-28s acc=%d threads=%4d blocks=%5d : %7.2f counter ticks per op (wave 0), kernel %.3f ms -> %.2f T op-lanes/s\n", name, NACC, threads, blocks, c / ops, ms,
           ops * threads * (double)blocks / (ms * 1e-3) / 1e12);
}

int main() {
    double *a, *o; long long *cyc;
    hipMalloc(&a, 128 * 8); hipMalloc(&o, 8 * 1024 * 1024); hipMalloc(&cyc, 8);
    std::vector<double> h(128); for (int i = 0; i < 128; i++) h[i] = (i < 64) ? 1.0 + i : 0.5 + 0.01 * (i - 64);
    hipMemcpy(a, h.data(), 128 * 8, hipMemcpyHostToDevice);
    k_semantics<<<1, 64>>>(o, a); hipDeviceSynchronize();
    std::vector<double> r(256); hipMemcpy(r.data(), o, 256 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) { const double e = h[(l & ~15) + 3] * h[64 + l]; if (r[l] != e) { if (bad < 4) printf("newbcast lane %d: got %g expect %g\n", l, r[l], e); bad++; } }
    printf("row_newbcast semantics: %s\n", bad ? "MISMATCH" : "ok (vdst += src0[lane k of the row] * src1)");
    printf("inactive source lane (even lanes off, k=2): odd lanes got"); for (int l = 1; l < 64; l += 16) printf(" %g (x[%d]*m=%g)", r[64 + l] - 1.0, (l & ~15) + 2, h[(l & ~15) + 2] * h[64 + l]); printf("\n");
    bad = 0; for (int l = 0; l < 64; l++) { const double e = h[l] + h[l ^ 16]; if (r[128 + l] != e) bad++; }
    printf("permlane16 pair sum (debug form): %s\n", bad ? "MISMATCH" : "ok");
    bad = 0; for (int l = 0; l < 64; l++) { const double e = h[l % 16] + h[l % 16 + 16] + h[l % 16 + 32] + h[l % 16 + 48]; if (fabs(r[192 + l] - e) > 1e-12) { if (bad < 4) printf("rowsum lane %d: got %g expect %g\n", l, r[192 + l], e); bad++; } }
    printf("four-row sum by swap16 + swap32 (selection-free): %s\n", bad ? "MISMATCH" : "ok");
    // one wave alone, then the CU filled to 3 waves per SIMD on every CU
    run<0, 1>("v_fma_f64 dependent", o, a, cyc, 64, 1);
    run<1, 1>("v_fmac_f64_dpp dependent", o, a, cyc, 64, 1);
    run<0, 2>("v_fma_f64", o, a, cyc, 64, 1);
    run<1, 2>("v_fmac_f64_dpp", o, a, cyc, 64, 1);
    run<0, 4>("v_fma_f64", o, a, cyc, 64, 1);
    run<1, 4>("v_fmac_f64_dpp", o, a, cyc, 64, 1);
    run<2, 4>("v_mov_b64_dpp + v_fma_f64", o, a, cyc, 64, 1);
    run<3, 4>("4-row sum (swap16+swap32)", o, a, cyc, 64, 1);
    run<0, 4>("v_fma_f64", o, a, cyc, 256, 256 * 3);
    run<1, 4>("v_fmac_f64_dpp", o, a, cyc, 256, 256 * 3);
    run<1, 2>("v_fmac_f64_dpp", o, a, cyc, 256, 256 * 3);
    run<2, 4>("v_mov_b64_dpp + v_fma_f64", o, a, cyc, 256, 256 * 3);
    run<3, 4>("4-row sum (swap16+swap32)", o, a, cyc, 256, 256 * 3);
    return 0;
}

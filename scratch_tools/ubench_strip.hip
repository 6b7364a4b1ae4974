// micro-benchmark of the DP row engine: cycles per row for C = 1,2,4,8 (one wave per SIMD vs. 4 waves per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../lofreq_amd/csrc/lfq_dp.hip"

template <int C>
__global__ void ub_kernel(int n_chunks, long long *cycles, double *sink)
{
    __shared__ LfqRow s_rows[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    LfqRow r; r.p = 1e-3 * (1 + (lane & 7)); r.q = 1.0 - r.p;
    s_rows[wave][lane] = r;
    __syncthreads();
    LfqStrip<C> S;
    lfq_strip_init<C>(S, true, 0);
    const int lt = 63;
    const double tflag = (lane == lt) ? 1.0 : 0.0;
    long long t0 = clock64();
    for (int ch = 0; ch < n_chunks; ch++) {
        if (lfq_strip_chunk<C>(S, s_rows[wave], ~0ull, false, nullptr, nullptr, false, nullptr, nullptr, tflag, true, lt, 1.0, 1e300)) break;
    }
    long long t1 = clock64();
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
    double acc = 0; for (int j = 0; j < C; j++) acc += S.v[j];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + S.e;
}

template <int C> void run(int blocks, int n_chunks)
{
    long long *d_c; double *d_s;
    hipMalloc(&d_c, blocks * 4 * sizeof(long long)); hipMalloc(&d_s, blocks * 256 * sizeof(double));
    for (int it = 0; it < 2; it++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(ub_kernel<C>, dim3(blocks), dim3(256), 0, 0, n_chunks, d_c, d_s);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<long long> h(blocks * 4);
        hipMemcpy(h.data(), d_c, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto x : h) avg += x; avg /= h.size();
        if (it == 1) printf("C=%d blocks=%d waves/SIMD=%.1f: %.1f clock64-ticks/row, %.3f ms, %.1f ns/row\n", C, blocks, blocks * 4 / 1024.0, avg / (64.0 * n_chunks), ms, ms * 1e6 / (64.0 * n_chunks));
    }
    hipFree(d_c); hipFree(d_s);
}

int main()
{
    for (int blocks : {256, 1024}) {
        run<1>(blocks, 157); run<2>(blocks, 157); run<4>(blocks, 157); run<8>(blocks, 157);
    }
    return 0;
}

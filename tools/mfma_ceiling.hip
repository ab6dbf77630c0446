// mfma_ceiling.hip — what the box actually sustains: (a) register-only MFMA loop, (b) MFMA fed by
// ds_read_b128 fragments at the conv kernels' ratios, for 1/2 waves per SIMD.  Calibrates the
// roofline fraction reported by bench.py.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k_reg(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f16v acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// PF pixel fragments x CF cout fragments per wave, each k-step reads PF + CF fragments from LDS.
template <int PF, int CF, int PITCH, int BAR>
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16v acc[PF][CF];
    for (int p = 0; p < PF; ++p) for (int c = 0; c < CF; ++c) for (int i = 0; i < 16; ++i) acc[p][c][i] = 0.f;
    const unsigned char* abase = lds + wave * 2048 + (lane & 31) * PITCH + (lane >> 5) * 16;
    const unsigned char* wbase = lds + 32768 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");  // LDS contents are "new" every chunk: no hoisting of the fragment reads
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            if (BAR > 0 && ks % BAR == 0) __syncthreads();
            h8 af[PF], wf[CF];
#pragma unroll
            for (int p = 0; p < PF; ++p) af[p] = *reinterpret_cast<const h8*>(abase + p * 32 * PITCH + (ks % 9) * PITCH + (ks / 9) * 32);
#pragma unroll
            for (int c = 0; c < CF; ++c) wf[c] = *reinterpret_cast<const h8*>(wbase + (ks * CF + c) * 1024);
#pragma unroll
            for (int p = 0; p < PF; ++p)
#pragma unroll
                for (int c = 0; c < CF; ++c) acc[p][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[p], wf[c], acc[p][c], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int p = 0; p < PF; ++p) for (int c = 0; c < CF; ++c) for (int i = 0; i < 16; ++i) s += acc[p][c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double timeit(F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5.0;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    float* out; CK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    const double fl = 2.0 * 32 * 32 * 16;
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int grid = cus * bpc, iters = 4000;
        double ms = timeit([&] { hipLaunchKernelGGL(k_reg<4>, dim3(grid), dim3(256), 0, 0, out, iters); });
        printf("{\"test\": \"reg_only_acc4\", \"waves_per_simd\": %d, \"tflops\": %.1f}\n", bpc, grid * 4.0 * iters * 4 * fl / (ms * 1e-3) / 1e12);
        ms = timeit([&] { hipLaunchKernelGGL(k_reg<1>, dim3(grid), dim3(256), 0, 0, out, iters); });
        printf("{\"test\": \"reg_only_acc1_dependent\", \"waves_per_simd\": %d, \"tflops\": %.1f}\n", bpc, grid * 4.0 * iters * 1 * fl / (ms * 1e-3) / 1e12);
    }
#define LDS_TEST(PF, CF, PITCH, BAR) \
    for (int bpc = 1; bpc <= 2; ++bpc) { \
        const int grid = cus * bpc, iters = 200; \
        CK(hipFuncSetAttribute((const void*)k_lds<PF, CF, PITCH, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); \
        double ms = timeit([&] { hipLaunchKernelGGL((k_lds<PF, CF, PITCH, BAR>), dim3(grid), dim3(256), 65536, 0, out, iters); }); \
        printf("{\"test\": \"lds_fed_p%dc%d_pitch%d_bar%d\", \"waves_per_simd\": %d, \"tflops\": %.1f}\n", PF, CF, PITCH, BAR, bpc, \
               grid * 4.0 * iters * 18.0 * PF * CF * fl / (ms * 1e-3) / 1e12); \
    }
    LDS_TEST(1, 1, 80, 0) LDS_TEST(1, 2, 80, 0) LDS_TEST(2, 1, 80, 0) LDS_TEST(2, 2, 80, 0) LDS_TEST(2, 2, 64, 0) LDS_TEST(2, 4, 80, 0)
    LDS_TEST(1, 1, 80, 2) LDS_TEST(2, 1, 80, 2) LDS_TEST(2, 2, 80, 2) LDS_TEST(1, 1, 80, 18) LDS_TEST(2, 1, 80, 18) LDS_TEST(2, 2, 80, 18)
    return 0;
}

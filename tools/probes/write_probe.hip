// write_probe.hip -- what HBM write rate does the macro tape's store pattern reach with no arithmetic in front of it?
// Build: hipcc --offload-arch=gfx950 -O3 -o write_probe write_probe.hip ; run on the GPU box.
// Pattern of macro_rollout_fwd*: grid = L workgroups, every step each workgroup stores one row of 2 x Nq float4
// ([step][lane][2][Nq][4] float32), rows of one step contiguous over the lanes, T steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int kMode>   // 0 plain, 1 nontemporal, 2 plain + a workgroup barrier per step
__global__ void tape_pattern(float4 *tape, int L, int Nq, int T, int spin) {
    const int lane = blockIdx.x, tid = threadIdx.x, B = blockDim.x;
    float acc = 0.f;
    for (int n = 0; n < T; ++n) {
        float4 *tp = tape + ((size_t)n * L + lane) * 2 * Nq;
        for (int k = 0; k < spin; ++k) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);     // stand-in for the solve
        for (int i = tid; i < 2 * Nq; i += B) {
            const float4 v = make_float4(acc, 1.f, (float)n, (float)i);
            if (kMode == 1) {
                __builtin_nontemporal_store(v.x, &tp[i].x); __builtin_nontemporal_store(v.y, &tp[i].y);
                __builtin_nontemporal_store(v.z, &tp[i].z); __builtin_nontemporal_store(v.w, &tp[i].w);
            } else tp[i] = v;
        }
        if (kMode == 2) __syncthreads();
    }
}
__global__ void fill_stride(float4 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void read_stride(const float4 *p, size_t n, float *out) {
    float a = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i]; a += v.x + v.y + v.z + v.w;
    }
    if (a == 123.456f) *out = a;
}

int main(int argc, char **argv) {
    const int L = 1024, Nq = 520, T = 1000;
    const size_t n4 = (size_t)T * L * 2 * Nq;
    float4 *tape; float *out;
    CK(hipMalloc(&tape, n4 * sizeof(float4))); CK(hipMalloc(&out, 4));
    CK(hipMemset(tape, 0, n4 * sizeof(float4)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](const char *name, auto fn, double bytes) {
        fn(); fn();
        CK(hipEventRecord(a));
        for (int i = 0; i < 5; ++i) fn();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
        printf("%-58s %8.3f ms  %6.2f TB/s\n", name, ms, bytes / ms / 1e9); fflush(stdout);
    };
    const double bytes = (double)n4 * 16;
    time("grid-stride float4 fill (2048 x 256)", [&] { fill_stride<<<2048, 256>>>(tape, n4); }, bytes);
    time("grid-stride float4 read (2048 x 256)", [&] { read_stride<<<2048, 256>>>(tape, n4, out); }, bytes);
    for (int B : {256, 512, 1024}) {
        char nm[128];
        snprintf(nm, sizeof nm, "tape pattern, plain stores, %d threads/lane", B);
        time(nm, [&] { tape_pattern<0><<<L, B>>>(tape, L, Nq, T, 0); }, bytes);
        snprintf(nm, sizeof nm, "tape pattern, nontemporal stores, %d threads/lane", B);
        time(nm, [&] { tape_pattern<1><<<L, B>>>(tape, L, Nq, T, 0); }, bytes);
        snprintf(nm, sizeof nm, "tape pattern, plain + barrier per step, %d threads/lane", B);
        time(nm, [&] { tape_pattern<2><<<L, B>>>(tape, L, Nq, T, 0); }, bytes);
    }
    for (int spin : {200, 400, 800, 1600}) {
        char nm[128];
        snprintf(nm, sizeof nm, "tape pattern, plain, 256 thr, %d dependent fma per step", spin);
        time(nm, [&] { tape_pattern<0><<<L, 256>>>(tape, L, Nq, T, spin); }, bytes);
    }
    return 0;
}

// store_probe.hip -- microbenchmark (not part of the product): what does the traversal kernel's
// memory pattern cost by itself?  1563 waves x 48 steps; per step each wave optionally does
// NFMA dependent-free fp64 FMAs per lane, optionally loads 8 KiB, and stores 8 KiB to slab[step].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NFMA, bool LOAD, bool STORE, bool SPREAD, bool NT = false>
__global__ __launch_bounds__(256, 2) void probe(double *const *slabs, int nsteps, long ntiles, double *out) {
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const size_t off = (size_t)tile * 1024 + lane * 2;  // doubles: tile*64*16 + lane*2
    double v[16];
    for (int e = 0; e < 16; e++) v[e] = 1.0 + lane * 1e-3 + e;
    for (int k = 0; k < nsteps; k++) {
        double *dst = slabs[k] + off;
        if (LOAD && k >= 2) {
            const double *src = slabs[k - 2] + off;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                double2 t = *reinterpret_cast<const double2 *>(src + j * 128);
                v[2 * j] += t.x; v[2 * j + 1] += t.y;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int r = 0; r < NFMA / 16; r++)
#pragma unroll
                for (int e = 0; e < 4; e++) v[c * 4 + e] = fma(v[c * 4 + e], 1.0000001, 1e-9);
            if (STORE && SPREAD) {
                *reinterpret_cast<double2 *>(dst + (2 * c) * 128) = make_double2(v[c * 4], v[c * 4 + 1]);
                *reinterpret_cast<double2 *>(dst + (2 * c + 1) * 128) = make_double2(v[c * 4 + 2], v[c * 4 + 3]);
            }
        }
        if (STORE && !SPREAD) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (NT) {
                    __builtin_nontemporal_store(v[2 * j], dst + j * 128);
                    __builtin_nontemporal_store(v[2 * j + 1], dst + j * 128 + 1);
                } else {
                    *reinterpret_cast<double2 *>(dst + j * 128) = make_double2(v[2 * j], v[2 * j + 1]);
                }
            }
        }
    }
    double s = 0;
    for (int e = 0; e < 16; e++) s += v[e];
    if (s == 12345.678) out[0] = s;
}

// Round 3: where do the 18 % between this pattern's 5.15 TB/s and a plain streaming store go?
// MODE 0: the traversal's pattern -- wave w writes its 8 KiB tile of slab[k], k = 0..nsteps-1 (eight 1 KiB instructions)
// MODE 1: the same bytes into ONE array laid out [step][tile] (no 48 separately allocated streams)
// MODE 2: one array laid out [tile][step]: a wave writes nsteps * 8 KiB contiguous bytes
// MODE 3: plain streaming store, grid-stride: instruction i of the whole grid writes the i-th KiB of one array
// MODE 4: as 0, but the eight instructions of a step go to eight DIFFERENT tiles' slices (slice j of tile (t + j) mod ntiles):
//         same bytes per step, no 8 KiB locality per wave
// MODE 5: as 0 with 2 KiB per instruction pair swapped in order (7,6,...,0): descending addresses inside the tile
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe2(double *const *slabs, double *one, int nsteps, long ntiles) {
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    double2 v = make_double2(1.0 + lane, 2.0 + lane);
    const long nwaves = (long)gridDim.x * 4;
    for (int k = 0; k < nsteps; k++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            double *dst;
            if (MODE == 0) dst = slabs[k] + (size_t)tile * 1024 + j * 128;
            else if (MODE == 1) dst = one + ((size_t)k * ntiles + tile) * 1024 + j * 128;
            else if (MODE == 2) dst = one + ((size_t)tile * nsteps + k) * 1024 + j * 128;
            else if (MODE == 3) dst = one + (((size_t)k * 8 + j) * nwaves + tile) * 128;
            else if (MODE == 4) dst = slabs[k] + (size_t)((tile + j * 97) % ntiles) * 1024 + j * 128;
            else dst = slabs[k] + (size_t)tile * 1024 + (7 - j) * 128;
            *reinterpret_cast<double2 *>(dst + lane * 2) = v;
        }
        v.x += 1.0;
    }
}

template <int MODE>
float run2(double *const *d_slabs, double *one, int nsteps, long ntiles, const char *name, double mb) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int grid = (int)((ntiles + 3) / 4);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((probe2<MODE>), dim3(grid), dim3(256), 0, 0, d_slabs, one, nsteps, ntiles);
    CHECK(hipEventRecord(a));
    const int reps = 20;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL((probe2<MODE>), dim3(grid), dim3(256), 0, 0, d_slabs, one, nsteps, ntiles);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    printf("%-72s %8.4f ms  %.2f TB/s\n", name, ms, mb / ms / 1e3);
    return ms;
}

template <int NFMA, bool LOAD, bool STORE, bool SPREAD, bool NT = false>
float run(double *const *d_slabs, int nsteps, long ntiles, double *d_out, const char *name) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int grid = (int)((ntiles + 3) / 4);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((probe<NFMA, LOAD, STORE, SPREAD, NT>), dim3(grid), dim3(256), 0, 0, d_slabs, nsteps, ntiles, d_out);
    CHECK(hipEventRecord(a));
    const int reps = 20;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL((probe<NFMA, LOAD, STORE, SPREAD, NT>), dim3(grid), dim3(256), 0, 0, d_slabs, nsteps, ntiles, d_out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    printf("%-44s %8.4f ms\n", name, ms);
    return ms;
}

int main(int argc, char **argv) {
    const long nptn = argc > 1 ? atol(argv[1]) : 100000;
    const int nsteps = 48;
    const long ntiles = (nptn + 63) / 64;
    std::vector<double *> slabs(nsteps);
    for (int k = 0; k < nsteps; k++) { CHECK(hipMalloc(&slabs[k], ntiles * 1024 * sizeof(double))); CHECK(hipMemset(slabs[k], 0, ntiles * 1024 * sizeof(double))); }
    double **d_slabs; CHECK(hipMalloc(&d_slabs, nsteps * sizeof(double *)));
    CHECK(hipMemcpy(d_slabs, slabs.data(), nsteps * sizeof(double *), hipMemcpyHostToDevice));
    double *d_out; CHECK(hipMalloc(&d_out, 8));
    const double mb = nsteps * ntiles * 8192.0 / 1e6;
    printf("patterns %ld, waves %ld, bytes stored per launch %.1f MB\n", nptn, ntiles, mb);
    {
        double *one; CHECK(hipMalloc(&one, (size_t)nsteps * (ntiles + 4) * 1024 * sizeof(double)));
        CHECK(hipMemset(one, 0, (size_t)nsteps * (ntiles + 4) * 1024 * sizeof(double)));
        run2<0>(d_slabs, one, nsteps, ntiles, "0: traversal pattern, 48 slabs, 8 KiB tile per wave and step", mb);
        run2<1>(d_slabs, one, nsteps, ntiles, "1: one array [step][tile]", mb);
        run2<2>(d_slabs, one, nsteps, ntiles, "2: one array [tile][step] (a wave writes 384 KiB contiguous)", mb);
        run2<3>(d_slabs, one, nsteps, ntiles, "3: plain streaming store (instruction i -> i-th KiB)", mb);
        run2<4>(d_slabs, one, nsteps, ntiles, "4: as 0, the 8 slices of a step to 8 different tiles", mb);
        run2<5>(d_slabs, one, nsteps, ntiles, "5: as 0, slices in descending address order", mb);
        CHECK(hipFree(one));
    }
    float t;
    t = run<0, false, true, false>(d_slabs, nsteps, ntiles, d_out, "stores only (end of step)"); printf("   -> %.2f TB/s\n", mb / t / 1e3);
    run<0, false, true, true>(d_slabs, nsteps, ntiles, d_out, "stores only (spread)");
    t = run<0, false, true, false, true>(d_slabs, nsteps, ntiles, d_out, "non-temporal stores only (end of step)"); printf("   -> %.2f TB/s\n", mb / t / 1e3);
    run<256, false, true, false, true>(d_slabs, nsteps, ntiles, d_out, "256 FMA + non-temporal stores at end");
    run<256, false, false, false>(d_slabs, nsteps, ntiles, d_out, "256 FMA/lane/step only");
    run<512, false, false, false>(d_slabs, nsteps, ntiles, d_out, "512 FMA/lane/step only");
    run<256, false, true, false>(d_slabs, nsteps, ntiles, d_out, "256 FMA + stores at end");
    run<256, false, true, true>(d_slabs, nsteps, ntiles, d_out, "256 FMA + stores spread");
    run<512, false, true, false>(d_slabs, nsteps, ntiles, d_out, "512 FMA + stores at end");
    run<512, false, true, true>(d_slabs, nsteps, ntiles, d_out, "512 FMA + stores spread");
    run<256, true, true, true>(d_slabs, nsteps, ntiles, d_out, "256 FMA + loads(k-2) + stores spread");
    run<512, true, true, true>(d_slabs, nsteps, ntiles, d_out, "512 FMA + loads(k-2) + stores spread");
    return 0;
}

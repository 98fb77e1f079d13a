/* A plain-C client of the C ABI (include/dhts.h): what a maintainer of a compiled host would write.  No torch, no C++: the HIP
 * runtime for device memory (hipMalloc / hipMemcpy through its C API), libdhts.so for the work, the C oracle as the checker.
 * One step of 3 lanes x 100 cells through dhts_macro_step_fwd / _bwd -- the drop-in for a batch of dMacroForwardLayer.forward /
 * .backward calls (reference road/lane/dmacro_lane.py:234-309) -- compared with oracle_macro_step / oracle_macro_step_bwd.
 * Build (tests/test_c_client.py does): gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Ioracle ... -ldhts -ldhts_oracle -lamdhip64 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "dhts.h"
#include "dhts_oracle.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_DHTS(x) do { int e_ = (x); if (e_ != DHTS_OK) { fprintf(stderr, "dhts status %d at %s:%d\n", e_, __FILE__, __LINE__); return 3; } } while (0)

static float frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xffffff) / 16777216.0f; }

int main(void) {
    enum { L = 3, N = 100 };
    const double dt = 0.01, dx = 5.0, um = 30.0;
    const int Np = dhts_padded(N);
    unsigned seed = 12345u;
    float r[L * N], u[L * N], y[L * N], q[L * N], ghost[L * 8], g_r[L * N], g_y[L * N];
    for (int i = 0; i < L * N; ++i) { r[i] = 0.05f + 0.9f * frand(&seed); u[i] = (float)um * frand(&seed); g_r[i] = frand(&seed) - 0.5f; g_y[i] = frand(&seed) - 0.5f; }
    /* y = r (u - u_eq(r)): FullQ.from_r_u (model/macro/_arz.py:73-86) in the reference's float32 glue, via the oracle */
    float gr[L * 2], gu[L * 2], gy[L * 2], gq[L * 2];
    for (int i = 0; i < L * N; ++i) oracle_arz_from_r_u(r[i], u[i], (float)um, &y[i], &q[i]);
    for (int i = 0; i < L * 2; ++i) { gr[i] = 0.05f + 0.9f * frand(&seed); gu[i] = (float)um * frand(&seed); oracle_arz_from_r_u(gr[i], gu[i], (float)um, &gy[i], &gq[i]); }
    for (int l = 0; l < L; ++l)
        for (int s = 0; s < 2; ++s) { float *g = ghost + (l * 2 + s) * 4; g[0] = gr[l * 2 + s]; g[1] = gy[l * 2 + s]; g[2] = gu[l * 2 + s]; g[3] = gq[l * 2 + s]; }

    /* ---- the oracle (CPU): a lane's four arrays carry the ghosts at index 0 and N + 1 ---- */
    float o_r[L * N], o_y[L * N], o_u[L * N], o_q[L * N], o_dqs[L * N * 12], o_gr[L * (N + 2)], o_gy[L * (N + 2)];
    for (int l = 0; l < L; ++l) {
        float pr[N + 2], py[N + 2], pu[N + 2], pq[N + 2];
        pr[0] = gr[l * 2]; py[0] = gy[l * 2]; pu[0] = gu[l * 2]; pq[0] = gq[l * 2];
        pr[N + 1] = gr[l * 2 + 1]; py[N + 1] = gy[l * 2 + 1]; pu[N + 1] = gu[l * 2 + 1]; pq[N + 1] = gq[l * 2 + 1];
        memcpy(pr + 1, r + l * N, sizeof(float) * N); memcpy(py + 1, y + l * N, sizeof(float) * N);
        memcpy(pu + 1, u + l * N, sizeof(float) * N); memcpy(pq + 1, q + l * N, sizeof(float) * N);
        int ei = -1;
        if (oracle_macro_step(N, pr, py, pu, pq, dt, dx, um, o_r + l * N, o_y + l * N, o_u + l * N, o_q + l * N, o_dqs + l * N * 12,
                              NULL, NULL, &ei) != 0) { fprintf(stderr, "oracle step failed\n"); return 4; }
        oracle_macro_step_bwd(N, o_dqs + l * N * 12, g_r + l * N, g_y + l * N, o_gr + l * (N + 2), o_gy + l * (N + 2));
    }

    /* ---- the device path ---- */
    dhts_macro_desc d = { L, N, dt, dx, um };
    const size_t tape_bytes = dhts_macro_step_tape_bytes(&d);
    if (tape_bytes != (size_t)L * 3 * Np * 16) { fprintf(stderr, "tape bytes %zu\n", tape_bytes); return 5; }
    float *d_r, *d_y, *d_u, *d_q, *d_g, *d_ro, *d_yo, *d_uo, *d_qo, *d_tape, *d_gr, *d_gy, *d_gro, *d_gyo;
    double *d_gg;
    dhts_error *d_err;
    const size_t sz = sizeof(float) * L * N;
    CHECK_HIP(hipMalloc((void **)&d_r, sz)); CHECK_HIP(hipMalloc((void **)&d_y, sz)); CHECK_HIP(hipMalloc((void **)&d_u, sz)); CHECK_HIP(hipMalloc((void **)&d_q, sz));
    CHECK_HIP(hipMalloc((void **)&d_ro, sz)); CHECK_HIP(hipMalloc((void **)&d_yo, sz)); CHECK_HIP(hipMalloc((void **)&d_uo, sz)); CHECK_HIP(hipMalloc((void **)&d_qo, sz));
    CHECK_HIP(hipMalloc((void **)&d_gr, sz)); CHECK_HIP(hipMalloc((void **)&d_gy, sz)); CHECK_HIP(hipMalloc((void **)&d_gro, sz)); CHECK_HIP(hipMalloc((void **)&d_gyo, sz));
    CHECK_HIP(hipMalloc((void **)&d_g, sizeof(ghost))); CHECK_HIP(hipMalloc((void **)&d_tape, tape_bytes));
    CHECK_HIP(hipMalloc((void **)&d_gg, sizeof(double) * L * 4)); CHECK_HIP(hipMalloc((void **)&d_err, sizeof(dhts_error)));
    CHECK_HIP(hipMemcpy(d_r, r, sz, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(d_y, y, sz, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_u, u, sz, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(d_q, q, sz, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_g, ghost, sizeof(ghost), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_gr, g_r, sz, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(d_gy, g_y, sz, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(d_err, 0, sizeof(dhts_error))); CHECK_HIP(hipMemset(d_gg, 0, sizeof(double) * L * 4));
    CHECK_DHTS(dhts_macro_step_fwd(&d, d_r, d_y, d_u, d_q, d_g, d_ro, d_yo, d_uo, d_qo, d_tape, d_err, NULL));
    CHECK_DHTS(dhts_macro_step_bwd(&d, d_tape, d_gr, d_gy, d_gro, d_gyo, d_gg, d_err, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float h_r[L * N], h_y[L * N], h_gr[L * N], h_gy[L * N];
    dhts_error herr;
    CHECK_HIP(hipMemcpy(h_r, d_ro, sz, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(h_y, d_yo, sz, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_gr, d_gro, sz, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(h_gy, d_gyo, sz, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&herr, d_err, sizeof(herr), hipMemcpyDeviceToHost));
    if (herr.code != DHTS_FAULT_NONE) { fprintf(stderr, "fault %d\n", herr.code); return 6; }

    /* ---- compare: next (r, y) bit for bit; cotangents of the lane's own cells to 1e-6 of their largest entry ---- */
    int bits = 0;
    double gmax = 0., gerr = 0.;
    for (int i = 0; i < L * N; ++i) bits += memcmp(&h_r[i], &o_r[i], 4) != 0 || memcmp(&h_y[i], &o_y[i], 4) != 0;
    for (int l = 0; l < L; ++l)
        for (int i = 0; i < N; ++i) {
            const double a = o_gr[l * (N + 2) + i + 1], b = o_gy[l * (N + 2) + i + 1];
            if (fabs(a) > gmax) gmax = fabs(a);
            if (fabs(b) > gmax) gmax = fabs(b);
            if (fabs(h_gr[l * N + i] - a) > gerr) gerr = fabs(h_gr[l * N + i] - a);
            if (fabs(h_gy[l * N + i] - b) > gerr) gerr = fabs(h_gy[l * N + i] - b);
        }
    printf("c client: %d of %d next-state entries differ from the oracle, cotangent error %.3g of %.3g\n", bits, 2 * L * N, gerr, gmax);
    return (bits == 0 && gerr <= 1e-6 * gmax) ? 0 : 1;
}

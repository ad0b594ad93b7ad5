// Symmetric eigensolver front end -- replaces numpy.linalg.eigh at utility_functions.py:58-59.
//
//   n <= 64 : two-sided cyclic Jacobi, one workgroup, A and V resident in LDS (the nx = 24 spatial problems of
//             GPCSD1D; also the leaf solver of the divide-and-conquer path).
//   n  > 64 : Householder tridiagonalisation + divide & conquer + compact-WY back-transformation (eigh_dc.hip).
// GPCSD_EIGH=jacobi in the environment forces the single-workgroup Jacobi on global memory for any n (slow; kept as
// an independent cross-check of the large-n solver).
#include <cstdlib>

#include "jacobi.hpp"
#include "kernels.hpp"

namespace gpcsd {

struct EigProb;   // eigh_dc.hip
void eigh_large_pair(gpcsd_ctx *c, double *A0, int n0, double *w0, double *Z0, const char *tag0, double *A1, int n1, double *w1,
                     double *Z1, const char *tag1, int *d_status, hipStream_t s);

template <int NT>
__global__ __launch_bounds__(NT) void jacobi_lds_kernel(const double *__restrict__ Ag, int n, double *evals, double *evecs,
                                                        int *status) {
    extern __shared__ double smem[];
    const int ld = n | 1;
    double *A = smem;
    double *V = A + n * ld;
    double *cs = V + n * ld;
    double *red = cs + 2 * (JACOBI_LDS_MAX / 2 + 1);
    __shared__ int pq[2 * JACOBI_LDS_MAX + 2];
    for (int e = threadIdx.x; e < n * n; e += NT) A[(e / n) * ld + (e % n)] = Ag[e];
    __syncthreads();
    jacobi_body<NT>(A, ld, V, ld, n, evals, evecs, n, status, cs, pq, red);
}

template <int NT>
__global__ __launch_bounds__(NT) void jacobi_global_kernel(double *A, double *V, int n, double *evals, double *evecs,
                                                           int *status) {
    __shared__ double cs[JACOBI_MAX_N + 2];
    __shared__ int pq[JACOBI_MAX_N + 2];
    __shared__ double red[JACOBI_MAX_N > NT ? JACOBI_MAX_N : NT];
    jacobi_body<NT>(A, n, V, n, n, evals, evecs, n, status, cs, pq, red);
}

static bool force_jacobi() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("GPCSD_EIGH");
        v = (e && std::string(e) == "jacobi") ? 1 : 0;
    }
    return v == 1;
}

static void eigh_jacobi(gpcsd_ctx *c, double *A, int n, double *evals, double *evecs, int *d_status, hipStream_t s,
                        const char *tag) {
    ProfScope ps(c, "eigh_jacobi", 9.0 * (double)n * n * n, s);
    if (n <= JACOBI_LDS_MAX) {
        const int ld = n | 1;
        const size_t sh = sizeof(double) * (2 * (size_t)n * ld + 2 * (JACOBI_LDS_MAX / 2 + 1) + 256);
        hipLaunchKernelGGL((jacobi_lds_kernel<256>), dim3(1), dim3(256), sh, s, (const double *)A, n, evals, evecs, d_status);
    } else {
        double *V = c->buf<double>(std::string("eigh_V_") + (tag ? tag : ""), (size_t)n * n);
        hipLaunchKernelGGL((jacobi_global_kernel<1024>), dim3(1), dim3(1024), 0, s, A, V, n, evals, evecs, d_status);
    }
    GP_HIP(hipGetLastError());
}

static void eigh_pair_enqueue(gpcsd_ctx *c, double *A0, int n0, double *w0, double *Z0, double *A1, int n1, double *w1,
                              double *Z1, int *d_status, hipStream_t s) {
    const bool small0 = n0 <= JACOBI_LDS_MAX || force_jacobi();
    const bool small1 = n1 <= JACOBI_LDS_MAX || force_jacobi();
    if (n0 > 0 && small0) eigh_jacobi(c, A0, n0, w0, Z0, d_status, s, "p0");
    if (n1 > 0 && small1) eigh_jacobi(c, A1, n1, w1, Z1, d_status, s, "p1");
    const bool big0 = n0 > 0 && !small0, big1 = n1 > 0 && !small1;
    if (big0 || big1)
        eigh_large_pair(c, big0 ? A0 : nullptr, big0 ? n0 : 0, w0, Z0, "p0", big1 ? A1 : nullptr, big1 ? n1 : 0, w1, Z1, "p1",
                        d_status, s);
}

static bool graphs_enabled() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("GPCSD_NO_GRAPH");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

// The large-n path is ~700 dependent launches with nothing decided on the host (deflation counts stay on the
// device), so it replays as a hipGraph: first call eager (allocates workspaces), second call captured, later calls
// replayed.  A graph is retired whenever any context buffer is (re)allocated, since it holds raw device pointers.
void eigh_pair_device(gpcsd_ctx *c, double *A0, int n0, double *w0, double *Z0, double *A1, int n1, double *w1, double *Z1,
                      int *d_status, hipStream_t s) {
    GP_REQUIRE(n0 <= JACOBI_MAX_N && n1 <= JACOBI_MAX_N, -3, "eigh: n=%d/%d exceeds %d", n0, n1, JACOBI_MAX_N);
    const bool any_large = !force_jacobi() && (n0 > JACOBI_LDS_MAX || n1 > JACOBI_LDS_MAX);
    if (!any_large || c->prof_on || !graphs_enabled()) {
        eigh_pair_enqueue(c, A0, n0, w0, Z0, A1, n1, w1, Z1, d_status, s);
        return;
    }
    char key[256];
    snprintf(key, sizeof(key), "eigh|%p|%d|%p|%p|%p|%d|%p|%p|%p|%p", (void *)A0, n0, (void *)w0, (void *)Z0, (void *)A1, n1,
             (void *)w1, (void *)Z1, (void *)d_status, (void *)s);
    gpcsd_ctx::GraphSlot &g = c->graphs[key];
    if (g.exec && g.epoch == c->alloc_epoch) {
        GP_HIP(hipGraphLaunch(g.exec, s));
        return;
    }
    if (g.seen_epoch == c->alloc_epoch) {          // allocations are stable since the last eager run: capture now
        if (g.exec) {
            (void)hipGraphExecDestroy(g.exec);
            g.exec = nullptr;
        }
        hipGraph_t graph = nullptr;
        GP_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        try {
            eigh_pair_enqueue(c, A0, n0, w0, Z0, A1, n1, w1, Z1, d_status, s);
        } catch (...) {
            (void)hipStreamEndCapture(s, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
        }
        GP_HIP(hipStreamEndCapture(s, &graph));
        if (g.seen_epoch == c->alloc_epoch) {
            GP_HIP(hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
            g.epoch = c->alloc_epoch;
        }
        (void)hipGraphDestroy(graph);
        if (g.exec && g.epoch == c->alloc_epoch) {
            GP_HIP(hipGraphLaunch(g.exec, s));
            return;
        }
    }
    eigh_pair_enqueue(c, A0, n0, w0, Z0, A1, n1, w1, Z1, d_status, s);
    g.seen_epoch = c->alloc_epoch;
}

void eigh_device(gpcsd_ctx *c, double *A, int n, double *evals, double *evecs, int *d_status, hipStream_t s,
                 const char *tag) {
    GP_REQUIRE(n >= 1 && n <= JACOBI_MAX_N, -3, "eigh: n=%d outside [1,%d]", n, JACOBI_MAX_N);
    (void)tag;
    eigh_pair_device(c, A, n, evals, evecs, nullptr, 0, nullptr, nullptr, d_status, s);
}

}  // namespace gpcsd

// ORACLE (test infrastructure only) — declarations shared between the oracle's .cpp files.
#ifndef VSO_INTERNAL_H
#define VSO_INTERNAL_H
#include <cstddef>
#include <cstdint>
namespace vso {
double pinned_hypot(double a, double b);
void jacobi_svd32f(float *At, size_t astep, float *W, float *Vt, size_t vstep, int m, int n, int n1);
extern thread_local int g_last_sweeps, g_last_visits, g_last_rotations;   // of the last jacobi_svd32f call
void svd32f_full(const float *A, int m, int n, float *w, float *u, float *vt);
void sincos_deg_pinned(float angle_deg, float *s_out, float *c_out);
}  // namespace vso
#endif

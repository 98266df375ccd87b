/* exact fused multiply-add for the emulator (numpy has none) */
#include <math.h>
void vfma(int n, const double *a, const double *b, const double *c, double *o)
{
    for (int i = 0; i < n; i++) o[i] = fma(a[i], b[i], c[i]);
}

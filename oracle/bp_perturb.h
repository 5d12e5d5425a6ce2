/* bp_perturb.h -- TEST INFRASTRUCTURE ONLY.  Force-included (-include) when building a second copy of ldpc_oracle.c in
 * which every exp()/log() of orc_bp() comes back rounded one ulp up, one ulp down or untouched at (pseudo-)random:
 * tests/test_oracle_golden.py uses it to show that BP hard decisions and iteration counts do not hinge on the last bit
 * of the transcendental functions (device ocml vs host glibc). */
#include <math.h>
#include <stdint.h>
#include <string.h>
static uint64_t pert_state = 88172645463325252ull;
static double pert_ulp(double x) {
    pert_state ^= pert_state << 13; pert_state ^= pert_state >> 7; pert_state ^= pert_state << 17;
    int r = (int)(pert_state % 3) - 1;
    if (r == 0 || !isfinite(x) || x == 0) return x;
    return nextafter(x, r > 0 ? INFINITY : -INFINITY);
}
static double pexp(double x) { return pert_ulp(exp(x)); }
static double plog(double x) { return pert_ulp(log(x)); }
#define ORC_BP_EXP pexp
#define ORC_BP_LOG plog

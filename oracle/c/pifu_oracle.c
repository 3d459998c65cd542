/* CPU oracle for the MonoPort / PIFu query hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (monoport_amd/) never does.  Parity status: PINNED for the query /
 * index path against golden vectors generated from the reference's own Python modules
 * (oracle/gen_golden.py -> tests/golden/).  Exports an fp32 build (orc_*_f32, the timed CPU
 * baseline: same arithmetic type as the reference) and an fp64 build (orc_*_f64, the
 * high-precision checker).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Built with -ffp-contract=off: every fused multiply-add of the reference's CPU path is written
 * as RFMA, everything else rounds after each operation (op-order parity with torch). */
#define REAL float
#define SUFFIX _f32
#define RFMA(a, b, c) fmaf((a), (b), (c))
#include "pifu_oracle_body.inc"
#undef REAL
#undef SUFFIX
#undef RFMA

#define REAL double
#define SUFFIX _f64
#define RFMA(a, b, c) fma((a), (b), (c))
#include "pifu_oracle_body.inc"
#undef REAL
#undef SUFFIX
#undef RFMA

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

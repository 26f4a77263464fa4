/* dl3_ops.c — C / OpenMP restatement of the operators of the DeepLabV3+ path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A third, independently written evaluation of the arithmetic the reference delegates to TensorFlow (next to the numpy
 * operators of oracle/dl3_oracle.py and the torch-autograd restatement oracle/torch_ref.py): plain loops, no BLAS, no
 * vendor library.  Two uses: (1) oracle/c_backend.py plugs these operators into dl3_oracle's graph so that the float64
 * numpy oracle also finishes at the sizes BASELINE.json quotes (512x512) — the full-size GPU parity tests then have TWO
 * float64 restatements to agree with; (2) bench.py times it (float32) as "CPU-A", the repo's own CPU restatement of the
 * step, beside the torch/oneDNN proxy.  PARITY UNPINNED like the other two: TensorFlow/Keras cannot be run
 * here (oracle/dl3_oracle.py header).
 * Only tests/, __graft_entry__ (build + smoke) and bench.py's cpu_baseline leg may load the library built from this.
 * Build: gcc -O3 -fopenmp -shared -fPIC dl3_ops.c -o ../_build/libdl3ops.so -lm   (oracle/c/Makefile) */
#include <math.h>
#include <omp.h>
#include <stdlib.h>

#define REAL double
#define FN(n) dl3ops_##n##_f64
#include "dl3_ops_impl.h"
#undef REAL
#undef FN

#define REAL float
#define FN(n) dl3ops_##n##_f32
#include "dl3_ops_impl.h"
#undef REAL
#undef FN

int dl3ops_version(void) { return 1; }
int dl3ops_max_threads(void) { return omp_get_max_threads(); }
void dl3ops_set_threads(int n) { omp_set_num_threads(n); }

/* CPU oracle, C restatement -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Build: make -C oracle   ->  oracle/_build/libctc_oracle.so
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it. */
#include <math.h>
#include <stdlib.h>

#define REAL float
#define SUF _f32
#define EXPF expf
#define LOGF logf
#include "ctc_oracle_impl.h"
#undef REAL
#undef SUF
#undef EXPF
#undef LOGF

#define REAL double
#define SUF _f64
#define EXPF exp
#define LOGF log
#include "ctc_oracle_impl.h"

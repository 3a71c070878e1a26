// Entry points whose HIP implementation has not landed yet: fail loudly, never fall back.
#include "vstab_internal.h"
extern "C" {
#ifndef HAVE_FIT
int vstab_sample_fit_batch(vstab_ctx*, const float*, int, int, int, int, int, vstab_fit_record*) { vstab_set_error("vstab_sample_fit_batch: not built"); return 99; }
#endif
#ifndef HAVE_TRAJ
int vstab_trajectory(vstab_ctx*, const double*, int, int, double, double, double, int, double*, double*) { vstab_set_error("vstab_trajectory: not built"); return 99; }
#endif
}

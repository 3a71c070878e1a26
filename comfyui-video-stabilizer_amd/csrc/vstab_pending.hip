// (no pending entry points: every symbol of include/vstab.h has a HIP implementation)
#include "vstab_internal.h"

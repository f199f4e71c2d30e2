/* Drop-in replacement for the reference's ocl_prefix_sum.h: everything is declared in
 * acmatch.h (layer 2).  Put include/compat first on the include path when
 * recompiling ocl_worker.c / ocl_aho_grep.c against libacmatch.so. */
#ifndef ACM_COMPAT_OCL_PREFIX_SUM_H_
#define ACM_COMPAT_OCL_PREFIX_SUM_H_
#include "../acmatch.h"
#endif

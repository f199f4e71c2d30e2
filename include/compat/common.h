/* Drop-in replacement for the reference's common.h: error and sizing macros
 * its callers use (common.h:12-20, :52-55, :64-71, :96-98), plus acmatch.h. */
#ifndef ACM_COMPAT_COMMON_H_
#define ACM_COMPAT_COMMON_H_
#include <errno.h>
#include <malloc.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "../acmatch.h"

#define ERRX(ret, str) do { fprintf(stderr, str "\n"); exit(ret); } while (0)
#define ERRXV(ret, fmt, ...) do { fprintf(stderr, fmt "\n", __VA_ARGS__); exit(ret); } while (0)
#define ERR(ret, str) do { fprintf(stderr, str ": %s\n", strerror(errno)); exit(ret); } while (0)
#define ERRV(ret, fmt, ...) \
	do { fprintf(stderr, fmt ": %s\n", __VA_ARGS__, strerror(errno)); exit(ret); } while (0)
#define DPRINTF(...)
#define DPRINTF_D(t)
#define DPRINTF_U(t)
#define DPRINTF_S(t)
#define CEILDIV(x, y) (((x) + ((y) - 1)) / (y))
#define ROUNDUP(x, y) (((x) + ((y) - 1)) & ~((y) - 1))
#define MALLOC(n) memalign(0x1000, n)
#define FREE(p) free(p)
#define LEN(a) (sizeof(a) / sizeof(*(a)))
#ifndef min
#define min(A, B) ((A) < (B) ? (A) : (B))
#endif
#ifndef max
#define max(A, B) ((A) > (B) ? (A) : (B))
#endif
#endif

/*
 * oracle/ref_harness.c -- thin accessors around the reference's own acsmx.c.
 *
 * TEST INFRASTRUCTURE ONLY (see the header of oracle/acref.c and DESIGN.md section 2).  This file is linked with
 * /root/reference/acsmx.c and utils.c, compiled where they lie (never copied)
 * by oracle/Makefile into oracle/_ref/libacsmx_ref.so.  It runs the
 * reference's acsm_new / acsm_add_pattern / acsm_compile unmodified and reads
 * the resulting acsm->state_table (public in acsmx.h:66-87).
 *
 * acsm_gen_state_table() is NOT called: it needs a live cl_context
 * (acsmx.c:618-623 exits when clCreateBuffer fails) and this image has no
 * OpenCL device.  Its 12-line serialisation loop (acsmx.c:640-658) is applied
 * here to the reference's in-memory state_table instead, so the table that
 * comes out is the reference's table cell for cell.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "acsmx.h"

acsm_t *
ref_new(void)
{
	return acsm_new();
}

void
ref_add_pattern(acsm_t *a, const unsigned char *pat, int n, int iid)
{
	acsm_add_pattern(a, (unsigned char *)pat, n, 0, 0, 0, NULL, iid);
}

void
ref_compile(acsm_t *a)
{
	acsm_compile(a);
}

/* number of states as acsm_get_states() reports it after
 * acsm_gen_state_table (acsmx.c:615): highest id + 1 */
int
ref_num_states(acsm_t *a)
{
	return a->num_states + 1;
}

int
ref_max_pattern_len(acsm_t *a)
{
	return acsm_get_max_pattern_size(a);
}

int
ref_num_patterns(acsm_t *a)
{
	return a->num_patterns;
}

/* the full DFA row of a state, straight from the reference's table */
void
ref_row(acsm_t *a, int state, int32_t *out256)
{
	memcpy(out256, a->state_table[state].next_state, 256 * sizeof(int32_t));
}

int
ref_fail(acsm_t *a, int state)
{
	return a->state_table[state].fail_state;
}

int
ref_head_index(acsm_t *a, int state)
{
	acsm_pattern_t *m = a->state_table[state].match_list;
	return m ? (int)m->index : -1;
}

int
ref_match_list(acsm_t *a, int state, int *out, int cap)
{
	acsm_pattern_t *m;
	int n = 0;
	for (m = a->state_table[state].match_list; m; m = m->next) {
		if (n < cap)
			out[n] = (int)m->index;
		n++;
	}
	return n;
}

/* acsmx.c:640-658 applied to the reference's state_table; plane-1 cells the
 * reference leaves uninitialised are zero */
void
ref_fill_table(acsm_t *a, int32_t *table)
{
	int i, j, n = a->num_states + 1;
	for (i = 0; i < n; i++) {
		for (j = 0; j < 256; j++) {
			int st = a->state_table[i].next_state[j];
			if (a->state_table[st].match_list) {
				table[(size_t)i * 512 + j] = -st;
				table[(size_t)i * 512 + 256 + j] =
				    (int32_t)a->state_table[st].match_list->index;
			} else {
				table[(size_t)i * 512 + j] = st;
				table[(size_t)i * 512 + 256 + j] = 0;
			}
		}
	}
}

/* serial walk over the reference's own state_table (App. B.1 semantics) */
size_t
ref_scan_serial(acsm_t *a, const unsigned char *text, size_t n,
    long init_state, uint32_t *out_pos, int32_t *out_pat, size_t cap,
    long *final_state)
{
	size_t k, m = 0;
	long state = init_state;
	for (k = 0; k < n; k++) {
		long nx = a->state_table[state].next_state[text[k]];
		acsm_pattern_t *ml = nx ? a->state_table[nx].match_list : NULL;
		if (ml) { /* -0 == 0: transitions into state 0 never flag */
			if (m < cap) {
				out_pos[m] = (uint32_t)k;
				out_pat[m] = (int32_t)ml->index;
			}
			m++;
		}
		state = nx;
	}
	if (final_state)
		*final_state = state;
	return m;
}

/* acsm_get_patterns_table (acsmx.c:677-735): export iid/n/next-chain */
int
ref_patterns_table(acsm_t *a, int *iid, int *n, int *next_chain)
{
	acsm_pattern_t *t = acsm_get_patterns_table(a);
	int i;
	if (!t)
		return -1;
	for (i = 0; i < a->num_patterns; i++) {
		iid[i] = t[i].iid;
		n[i] = t[i].n;
		next_chain[i] = t[i].next ? (int)t[i].next->index : -1;
	}
	return a->num_patterns;
}

/* utils.c:32-54, the reference's own hex decoder (exits on odd length) */
extern unsigned char *printable_hex_to_bytes(unsigned char *);

int
ref_hex_to_bytes(const char *hex, unsigned char *out, int cap)
{
	int n = (int)(strlen(hex) / 2);
	unsigned char *b = printable_hex_to_bytes((unsigned char *)hex);
	memcpy(out, b, (size_t)(n < cap ? n : cap));
	free(b);
	return n;
}

void
ref_free(acsm_t *a)
{
	acsm_cleanup(a);
	acsm_free(a);
}

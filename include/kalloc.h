/*
 * kalloc.h — host allocator ABI that mwf_rst_t::cigar ownership is defined against.
 *
 * lh3/miniwfa hands every result buffer to the caller through its `void *km` arena handle
 * (reference kalloc.h:14-24; miniwfa.c:434 relocates r->cigar into the caller's km), and users
 * of the reference compile kalloc.c next to miniwfa.c (README.md:9,26).  libmwf_hip.so therefore
 * exports the same ten entry points with the same contracts:
 *
 *   km == NULL            -> plain libc malloc/calloc/realloc/free
 *   km_init()/km_init2()  -> an arena; km_init2(parent, n) draws its memory from `parent`
 *   km_destroy(km)        -> releases everything the arena ever handed out
 *
 * The implementation (miniwfa_amd/csrc/kalloc.cpp) is an address-ordered first-fit free list
 * with boundary coalescing, written from scratch.  An arena is not thread-safe; use one per thread
 * (same rule as the reference).  Device scratch never comes from here — the engine owns
 * per-stream hipMalloc pools.
 */
#ifndef MWF_HIP_KALLOC_H
#define MWF_HIP_KALLOC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { /* reference kalloc.h:10-12 */
	size_t capacity, available, n_blocks, n_cores, largest;
} km_stat_t;

void *kmalloc(void *km, size_t size);                 /* size==0 -> NULL */
void *kcalloc(void *km, size_t count, size_t size);   /* zero-filled */
void *krealloc(void *km, void *ptr, size_t size);     /* size==0 frees and returns NULL; never shrinks in place */
void *krelocate(void *km, void *ap, size_t n_bytes);  /* km==NULL: returns ap; else a compact copy inside km, ap freed */
void  kfree(void *km, void *ptr);

void *km_init(void);
void *km_init2(void *km_par, size_t min_core_size);
void  km_destroy(void *km);
void  km_stat(const void *km, km_stat_t *s);
void  km_stat_print(const void *km);

#ifdef __cplusplus
}
#endif

#define Kmalloc(km, type, cnt)       ((type*)kmalloc((km), (cnt) * sizeof(type)))
#define Kcalloc(km, type, cnt)       ((type*)kcalloc((km), (cnt), sizeof(type)))
#define Krealloc(km, type, ptr, cnt) ((type*)krealloc((km), (ptr), (cnt) * sizeof(type)))

#endif

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

/* typed convenience forms, reference kalloc.h:29-31 */
#define Kmalloc(km, type, cnt)       ((type*)kmalloc((km), (cnt) * sizeof(type)))
#define Kcalloc(km, type, cnt)       ((type*)kcalloc((km), (cnt), sizeof(type)))
#define Krealloc(km, type, ptr, cnt) ((type*)krealloc((km), (ptr), (cnt) * sizeof(type)))

/* assign-in-place forms, reference kalloc.h:33-35: `ptr` receives `len` elements of its own pointee type */
#define KMALLOC(km, ptr, len)  ((ptr) = (__typeof__(ptr))kmalloc((km), (len) * sizeof(*(ptr))))
#define KCALLOC(km, ptr, len)  ((ptr) = (__typeof__(ptr))kcalloc((km), (len), sizeof(*(ptr))))
#define KREALLOC(km, ptr, len) ((ptr) = (__typeof__(ptr))krealloc((km), (ptr), (len) * sizeof(*(ptr))))

/* grow array `a` of capacity `m` (an lvalue): 16 elements at first, then by half — reference kalloc.h:37-40 */
#define KEXPAND(km, a, m) \
	do { \
		(m) = (m) >= 4 ? (m) + ((m) >> 1) : 16; \
		KREALLOC((km), (a), (m)); \
	} while (0)

#ifndef klib_unused
#if defined(__GNUC__) || defined(__clang__)
#define klib_unused __attribute__((__unused__))
#else
#define klib_unused
#endif
#endif

/* Object pool over an arena, reference kalloc.h:50-80.  KALLOC_POOL_INIT(name, T) defines kmp_<name>_t and
 * kmp_init_<name>(km), kmp_destroy_<name>(mp), kmp_alloc_<name>(mp) (zero-filled when fresh, recycled as is),
 * kmp_free_<name>(mp, p) (parks p for re-use); `cnt` counts objects currently handed out. */
#define KALLOC_POOL_INIT2(SCOPE, name, kmptype_t) \
	typedef struct { \
		size_t cnt, n, max; \
		kmptype_t **buf; \
		void *km; \
	} kmp_##name##_t; \
	SCOPE kmp_##name##_t *kmp_init_##name(void *km) \
	{ \
		kmp_##name##_t *pool = (kmp_##name##_t*)kcalloc(km, 1, sizeof(kmp_##name##_t)); \
		pool->km = km; \
		return pool; \
	} \
	SCOPE void kmp_destroy_##name(kmp_##name##_t *pool) \
	{ \
		size_t i_; \
		for (i_ = 0; i_ < pool->n; ++i_) kfree(pool->km, pool->buf[i_]); \
		kfree(pool->km, pool->buf); \
		kfree(pool->km, pool); \
	} \
	SCOPE kmptype_t *kmp_alloc_##name(kmp_##name##_t *pool) \
	{ \
		++pool->cnt; \
		return pool->n ? pool->buf[--pool->n] : (kmptype_t*)kcalloc(pool->km, 1, sizeof(kmptype_t)); \
	} \
	SCOPE void kmp_free_##name(kmp_##name##_t *pool, kmptype_t *obj) \
	{ \
		--pool->cnt; \
		if (pool->n == pool->max) KEXPAND(pool->km, pool->buf, pool->max); \
		pool->buf[pool->n++] = obj; \
	}

#define KALLOC_POOL_INIT(name, kmptype_t) KALLOC_POOL_INIT2(static inline klib_unused, name, kmptype_t)

#endif

// kalloc.cpp — host arena allocator behind include/kalloc.h (ABI of reference kalloc.h:14-24).
//
// Written from scratch: an arena owns a list of "cores" (big chunks obtained from its parent
// arena, or from libc when it has no parent) and an address-ordered singly linked list of free
// blocks.  Allocation is first-fit with splitting from the block's tail; freeing re-inserts in
// address order and merges with both neighbours.  Every live block is preceded by one size_t
// holding its total size in bytes (header included).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "kalloc.h"

namespace {

struct FreeBlock {
	size_t bytes;      // whole block, header included
	FreeBlock *next;   // next free block by address (may live in another core)
};

struct Core {
	Core *next;
	size_t bytes;      // whole core, this header included
};

struct Arena {
	void *parent;
	size_t min_core;   // bytes
	FreeBlock *free_head;
	Core *cores;
};

constexpr size_t kAlign = 16;
constexpr size_t kHdr = 16; // keeps payloads 16-byte aligned; only the first size_t is used

[[noreturn]] void km_panic(const char *msg)
{
	fprintf(stderr, "[kalloc] %s\n", msg);
	abort();
}

inline size_t round_up(size_t n, size_t a) { return (n + a - 1) / a * a; }

void insert_free(Arena *a, FreeBlock *b)
{
	FreeBlock *prev = nullptr, *cur = a->free_head;
	while (cur && cur < b) prev = cur, cur = cur->next;
	if (prev && (char*)prev + prev->bytes > (char*)b) km_panic("kfree: block overlaps the free block before it (double free?)");
	if (cur && (char*)b + b->bytes > (char*)cur) km_panic("kfree: block overlaps the free block after it (double free?)");
	b->next = cur;
	if (cur && (char*)b + b->bytes == (char*)cur) { // merge forward
		b->bytes += cur->bytes;
		b->next = cur->next;
	}
	if (prev && (char*)prev + prev->bytes == (char*)b) { // merge backward
		prev->bytes += b->bytes;
		prev->next = b->next;
	} else if (prev) prev->next = b;
	else a->free_head = b;
}

void add_core(Arena *a, size_t need)
{
	size_t bytes = round_up(need + round_up(sizeof(Core), kAlign), a->min_core);
	Core *c = (Core*)kmalloc(a->parent, bytes);
	if (!c) km_panic("out of memory while growing an arena");
	c->next = a->cores, c->bytes = bytes, a->cores = c;
	FreeBlock *b = (FreeBlock*)((char*)c + round_up(sizeof(Core), kAlign));
	b->bytes = bytes - round_up(sizeof(Core), kAlign);
	b->next = nullptr;
	insert_free(a, b);
}

} // namespace

extern "C" {

void *km_init2(void *km_par, size_t min_core_size)
{
	Arena *a = (Arena*)kcalloc(km_par, 1, sizeof(Arena));
	a->parent = km_par;
	if (min_core_size > 0) a->min_core = round_up(min_core_size, kAlign);
	else if (km_par) a->min_core = ((Arena*)km_par)->min_core / 2 > 4096 ? ((Arena*)km_par)->min_core / 2 : 4096;
	else a->min_core = (size_t)8 << 20;
	return a;
}

void *km_init(void) { return km_init2(nullptr, 0); }

void km_destroy(void *km)
{
	Arena *a = (Arena*)km;
	if (!a) return;
	void *par = a->parent;
	for (Core *c = a->cores; c;) {
		Core *n = c->next;
		kfree(par, c);
		c = n;
	}
	kfree(par, a);
}

void *kmalloc(void *km, size_t size)
{
	if (size == 0) return nullptr;
	if (!km) return malloc(size);
	Arena *a = (Arena*)km;
	size_t need = round_up(size + kHdr, kAlign);
	if (need < sizeof(FreeBlock) + kHdr) need = round_up(sizeof(FreeBlock) + kHdr, kAlign);
	for (int attempt = 0; attempt < 2; ++attempt) {
		FreeBlock *prev = nullptr;
		for (FreeBlock *b = a->free_head; b; prev = b, b = b->next) {
			if (b->bytes < need) continue;
			char *blk;
			if (b->bytes - need >= round_up(sizeof(FreeBlock) + kHdr, kAlign)) { // split: hand out the tail
				b->bytes -= need;
				blk = (char*)b + b->bytes;
			} else { // take the whole block
				need = b->bytes;
				if (prev) prev->next = b->next; else a->free_head = b->next;
				blk = (char*)b;
			}
			*(size_t*)blk = need;
			return blk + kHdr;
		}
		add_core(a, need);
	}
	km_panic("kmalloc: no block found after growing the arena");
}

void *kcalloc(void *km, size_t count, size_t size)
{
	if (count == 0 || size == 0) return nullptr;
	if (!km) return calloc(count, size);
	void *p = kmalloc(km, count * size);
	memset(p, 0, count * size);
	return p;
}

void kfree(void *km, void *ptr)
{
	if (!ptr) return;
	if (!km) { free(ptr); return; }
	FreeBlock *b = (FreeBlock*)((char*)ptr - kHdr);
	size_t bytes = *(size_t*)b;
	b->bytes = bytes;
	insert_free((Arena*)km, b);
}

void *krealloc(void *km, void *ptr, size_t size)
{
	if (size == 0) { kfree(km, ptr); return nullptr; }
	if (!km) return realloc(ptr, size);
	if (!ptr) return kmalloc(km, size);
	size_t cap = *(size_t*)((char*)ptr - kHdr) - kHdr;
	if (cap >= size) return ptr;
	void *q = kmalloc(km, size);
	memcpy(q, ptr, cap);
	kfree(km, ptr);
	return q;
}

void *krelocate(void *km, void *ap, size_t n_bytes)
{
	if (!km || !ap) return ap;
	if (n_bytes == 0) { kfree(km, ap); return nullptr; }
	void *p = kmalloc(km, n_bytes);
	memcpy(p, ap, n_bytes);
	kfree(km, ap);
	return p;
}

void km_stat(const void *km, km_stat_t *s)
{
	memset(s, 0, sizeof(*s));
	const Arena *a = (const Arena*)km;
	if (!a) return;
	for (const FreeBlock *b = a->free_head; b; b = b->next) {
		s->available += b->bytes;
		++s->n_blocks;
	}
	for (const Core *c = a->cores; c; c = c->next) {
		++s->n_cores;
		s->capacity += c->bytes;
		if (c->bytes > s->largest) s->largest = c->bytes;
	}
}

void km_stat_print(const void *km)
{
	km_stat_t st;
	km_stat(km, &st);
	fprintf(stderr, "[km_stat] cap=%zu, avail=%zu, largest=%zu, n_core=%zu, n_block=%zu\n",
	        st.capacity, st.available, st.largest, st.n_cores, st.n_blocks);
}

} // extern "C"

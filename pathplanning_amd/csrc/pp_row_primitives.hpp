// Row-level (16 lanes = one DPP row) cross-lane primitives and the open-list operations built on them; used by
// k_hybrid_search_rows (pp_planner_rows.hpp), where one query owns one row of a wave.  Every function expects to be
// called with whole rows active (control flow may diverge between rows, never inside one).
#pragma once
#include "pp_search_device.hpp"

namespace ppd {

#ifndef PP_WAVE_SYNC_DRAIN
#define PP_WAVE_SYNC_DRAIN 1 // 0 (experiment): rely on the in-order execution of one wave's memory instructions instead of draining its stores
#endif
/// this wave's global stores are visible to its other lanes' loads
PPD_INLINE void row_vmem_sync()
{
#if PP_WAVE_SYNC_DRAIN
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	__builtin_amdgcn_s_waitcnt(0);
#else
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#endif
	__builtin_amdgcn_wave_barrier();
}

constexpr int kRowLanes = 16;
constexpr int kRowsPerWave = 4;
constexpr int kRowSlots = kRowLanes + 1; // staging per row: one slot per lane + the Reeds-Shepp child
constexpr int kRowRs = kRowLanes;

// ------------------------------------------------------------------------------------------ row primitives --
PPD_INLINE uint32_t row_bits(unsigned long long ballot, int lane) { return (uint32_t)(ballot >> (lane & 48)) & 0xFFFFu; }
/// value of lane c (0..15, uniform within the row) of the caller's row
PPD_INLINE uint32_t row_read(uint32_t v, int lane, int c) { return (uint32_t)__builtin_amdgcn_ds_bpermute(((lane & 48) | c) << 2, (int)v); }
PPD_INLINE unsigned long long row_read64(unsigned long long v, int lane, int c)
{
	return ((unsigned long long)row_read((uint32_t)(v >> 32), lane, c) << 32) | row_read((uint32_t)v, lane, c);
}
PPD_INLINE double row_read_f64(double v, int lane, int c) { return __longlong_as_double((long long)row_read64((unsigned long long)__double_as_longlong(v), lane, c)); }
/// lane i receives lane i-1 of its row; lane 0 of the row receives `fill`  (DPP row_shr:1)
PPD_INLINE uint32_t row_shr1(uint32_t v, uint32_t fill) { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x111, 0xF, 0xF, false); }
/// lane i receives lane i+1 of its row; lane 15 of the row receives `fill`  (DPP row_shl:1)
PPD_INLINE uint32_t row_shl1(uint32_t v, uint32_t fill) { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x101, 0xF, 0xF, false); }
PPD_INLINE unsigned long long row_shr1_64(unsigned long long v, unsigned long long fill)
{
	return ((unsigned long long)row_shr1((uint32_t)(v >> 32), (uint32_t)(fill >> 32)) << 32) | row_shr1((uint32_t)v, (uint32_t)fill);
}
PPD_INLINE unsigned long long row_shl1_64(unsigned long long v, unsigned long long fill)
{
	return ((unsigned long long)row_shl1((uint32_t)(v >> 32), (uint32_t)(fill >> 32)) << 32) | row_shl1((uint32_t)v, (uint32_t)fill);
}
/// minimum over the 16 lanes of the row, result in every lane: xor-butterfly with quad_perm / row_half_mirror / row_mirror
PPD_INLINE uint32_t row_min_u32(uint32_t v)
{
	uint32_t x = v;
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xF, 0xF, false)); // row_half_mirror
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xF, 0xF, false)); // row_mirror
	return x;
}
PPD_INLINE long long row_sum_i64(long long v, int lane)
{
	long long x = v;
#pragma unroll
	for (int off = 8; off > 0; off >>= 1)
		x += __shfl_xor(x, off, 64); // partners stay inside the row (off < 16)
	return x;
}
/// lane (0..15) of the lexicographically smallest (k, s) in the row; also returns that key in every lane
PPD_INLINE int row_argmin_key(unsigned long long k, uint32_t s, int lane, unsigned long long& mk, uint32_t& ms)
{
	const uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
	const uint32_t mhi = row_min_u32(hi);
	bool cand = hi == mhi;
	const uint32_t mlo = row_min_u32(cand ? lo : 0xFFFFFFFFu);
	cand = cand && lo == mlo;
	ms = row_min_u32(cand ? s : 0xFFFFFFFFu);
	cand = cand && s == ms;
	mk = ((unsigned long long)mhi << 32) | mlo;
	return __ffs((int)row_bits(__ballot(cand), lane)) - 1;
}
/// true in lane rl iff an EARLIER lane of the row (source index < rl) holds the same key with its flag set
PPD_INLINE bool row_earlier_same(uint32_t key, bool flag, int rl)
{
	const uint32_t f = flag ? 1u : 0u;
	bool dup = false;
#define PP_ROR(n)                                                                                                    \
	{                                                                                                                \
		const uint32_t k_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x120 + (n), 0xF, 0xF, false);         \
		const uint32_t f_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x120 + (n), 0xF, 0xF, false);           \
		dup = dup || ((n) <= rl && f_ != 0u && k_ == key); /* row_ror:n -> lane i reads lane (i - n) mod 16 */       \
	}
	PP_ROR(1) PP_ROR(2) PP_ROR(3) PP_ROR(4) PP_ROR(5) PP_ROR(6) PP_ROR(7) PP_ROR(8) PP_ROR(9) PP_ROR(10) PP_ROR(11) PP_ROR(12) PP_ROR(13) PP_ROR(14) PP_ROR(15)
#undef PP_ROR
	return dup;
}

// ---------------------------------------------------------------------------------- open list, row flavour --
/// Inserts e (uniform within the row).  Returns true when an entry left the 16-entry buffer (-> `spilled`).
PPD_INLINE bool front_insert_row(FrontLane& f, int& count, const HeapEntry& e, int rl, int lane, HeapEntry& spilled)
{
	const bool mineFirst = rl < count && key_before(f.ckey, f.nseq, e.ckey, e.nseq);
	const int pos = __popc(row_bits(__ballot(mineFirst), lane));
	if (pos >= kRowLanes) {
		spilled = e;
		return true;
	}
	bool spill = false;
	if (count == kRowLanes) {
		spilled.ckey = row_read64(f.ckey, lane, kRowLanes - 1);
		spilled.nseq = row_read(f.nseq, lane, kRowLanes - 1);
		spilled.node = row_read(f.node, lane, kRowLanes - 1);
		spill = true;
	}
	const unsigned long long uk = row_shr1_64(f.ckey, ~0ull);
	const uint32_t us = row_shr1(f.nseq, ~0u);
	const uint32_t un = row_shr1(f.node, 0u);
	if (rl > pos) {
		f.ckey = uk;
		f.nseq = us;
		f.node = un;
	} else if (rl == pos) {
		f.ckey = e.ckey;
		f.nseq = e.nseq;
		f.node = e.node;
	}
	if (count < kRowLanes)
		count++;
	return spill;
}
/// Removes and returns front[0] (count > 0).
PPD_INLINE HeapEntry front_pop_row(FrontLane& f, int& count, int lane)
{
	HeapEntry top;
	top.ckey = row_read64(f.ckey, lane, 0);
	top.nseq = row_read(f.nseq, lane, 0);
	top.node = row_read(f.node, lane, 0);
	f.ckey = row_shl1_64(f.ckey, ~0ull);
	f.nseq = row_shl1(f.nseq, ~0u);
	f.node = row_shl1(f.node, 0u);
	count--;
	return top;
}
/// sorts the 16 entries of a row ascending in (ckey, nseq): bitonic network, partners stay inside the row; keys are unique
PPD_INLINE void row_sort_entries(unsigned long long& ckey, uint32_t& nseq, uint32_t& node, int rl)
{
#pragma unroll
	for (int k = 2; k <= kRowLanes; k <<= 1) {
#pragma unroll
		for (int j = k >> 1; j > 0; j >>= 1) {
			const unsigned long long ok = ((unsigned long long)(uint32_t)__shfl_xor((int)(ckey >> 32), j, 64) << 32) | (uint32_t)__shfl_xor((int)ckey, j, 64);
			const uint32_t os = (uint32_t)__shfl_xor((int)nseq, j, 64), on = (uint32_t)__shfl_xor((int)node, j, 64);
			const bool up = (rl & k) == 0, lower = (rl & j) == 0;
			const bool otherBefore = ok < ckey || (ok == ckey && os < nseq);
			if ((lower == up) == otherBefore) {
				ckey = ok;
				nseq = os;
				node = on;
			}
		}
	}
}
/// 64-ary heap pop by the 16 lanes of a row (4 children per lane and level); see heap_pop_wave
PPD_INLINE HeapEntry heap_pop_row(HeapEntry* heap, int& size, int rl, int lane, HeapEntry& cachedTop)
{
	const HeapEntry top = cachedTop;
	const int hs = size - 1;
	size = hs;
	if (hs <= 0) {
		cachedTop.ckey = ~0ull;
		cachedTop.nseq = ~0u;
		cachedTop.node = 0;
		return top;
	}
	const HeapEntry last = heap[hs];
	HeapEntry newRoot = last;
	int i = 0;
	for (;;) {
		const int first = (i << 6) + 1;
		if (first >= hs)
			break;
		HeapEntry best;
		best.ckey = ~0ull;
		best.nseq = ~0u;
		best.node = 0;
		int bestIdx = first;
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const int c = first + rl + kRowLanes * k;
			if (c < hs) {
				const HeapEntry e = heap[c];
				if (heap_before(e, best)) {
					best = e;
					bestIdx = c;
				}
			}
		}
		unsigned long long mk;
		uint32_t ms;
		const int minLane = row_argmin_key(best.ckey, best.nseq, lane, mk, ms);
		HeapEntry rowBest;
		rowBest.ckey = mk;
		rowBest.nseq = ms;
		rowBest.node = 0;
		if (!heap_before(rowBest, last))
			break;
		const int minIdx = (int)row_read((uint32_t)bestIdx, lane, minLane);
		if (rl == minLane)
			heap[i] = best;
		if (i == 0) {
			rowBest.node = row_read(best.node, lane, minLane);
			newRoot = rowBest;
		}
		i = minIdx;
	}
	if (rl == 0)
		heap[i] = last;
	cachedTop = newRoot;
	return top;
}

/// std::mt19937_64 regeneration by the 16 lanes of a row, state in HBM (every 312 draws)
PPD_INLINE void mt_twist_row(unsigned long long* mt, int rl)
{
	const unsigned long long UM = 0xFFFFFFFF80000000ull, LM = 0x7FFFFFFFull, Acoef = 0xB5026F5AA96619E9ull;
	constexpr int N = Mt64::N, M = Mt64::M;
	for (int base = 0; base < M; base += kRowLanes) { // i in [0, 156): inputs are all old values
		const int i = base + rl;
		unsigned long long v = 0;
		if (i < M) {
			const unsigned long long x = (mt[i] & UM) | (mt[i + 1] & LM);
			v = mt[i + M] ^ (x >> 1) ^ ((x & 1ull) ? Acoef : 0ull);
		}
		row_vmem_sync();
		if (i < M)
			mt[i] = v;
		row_vmem_sync();
	}
	for (int base = M; base < N - 1; base += kRowLanes) { // i in [156, 311): mt[i - 156] is already new, mt[i], mt[i+1] old
		const int i = base + rl;
		unsigned long long v = 0;
		if (i < N - 1) {
			const unsigned long long x = (mt[i] & UM) | (mt[i + 1] & LM);
			v = mt[i - M] ^ (x >> 1) ^ ((x & 1ull) ? Acoef : 0ull);
		}
		row_vmem_sync();
		if (i < N - 1)
			mt[i] = v;
		row_vmem_sync();
	}
	if (rl == 0) {
		const unsigned long long x = (mt[N - 1] & UM) | (mt[0] & LM);
		mt[N - 1] = mt[M - 1] ^ (x >> 1) ^ ((x & 1ull) ? Acoef : 0ull);
	}
	row_vmem_sync();
}


} // namespace ppd

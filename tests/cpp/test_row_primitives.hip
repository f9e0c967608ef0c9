// GPU self-test of the DPP / bpermute row primitives used by k_hybrid_search_rows (pp_row_primitives.hpp).
// Build: hipcc --offload-arch=gfx950 -O2 -I pathplanning_amd/csrc -I include tests/cpp/test_row_primitives.hip -o pathplanning_amd/lib/test_row_primitives
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "pp_row_primitives.hpp"

using namespace ppd;

struct Out {
	uint32_t shr, shl, mn, rd, dup, argLane;
	unsigned long long argKey;
	uint32_t argSeq;
};

__global__ void k_test(const uint32_t* in, const uint32_t* keys, const uint32_t* flags, Out* out)
{
	const int lane = threadIdx.x, rl = lane & 15;
	const uint32_t v = in[lane];
	Out o;
	o.shr = row_shr1(v, 0xAAAAu);
	o.shl = row_shl1(v, 0xBBBBu);
	o.mn = row_min_u32(v);
	o.rd = row_read(v, lane, (rl * 7 + 3) & 15);
	o.dup = row_earlier_same(keys[lane], flags[lane] != 0, rl) ? 1u : 0u;
	unsigned long long mk;
	uint32_t ms;
	o.argLane = (uint32_t)row_argmin_key(((unsigned long long)keys[lane] << 32) | (v & 3u), v, lane, mk, ms);
	o.argKey = mk;
	o.argSeq = ms;
	out[lane] = o;
}

int main()
{
	std::vector<uint32_t> in(64), keys(64), flags(64);
	uint32_t s = 12345;
	auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
	int bad = 0;
	uint32_t *din, *dk, *df;
	Out* dout;
	hipMalloc(&din, 256);
	hipMalloc(&dk, 256);
	hipMalloc(&df, 256);
	hipMalloc(&dout, 64 * sizeof(Out));
	for (int trial = 0; trial < 50; trial++) {
		for (int i = 0; i < 64; i++) {
			in[i] = rnd() & 0xFFFF;
			keys[i] = rnd() % 5;
			flags[i] = rnd() & 1;
		}
		hipMemcpy(din, in.data(), 256, hipMemcpyHostToDevice);
		hipMemcpy(dk, keys.data(), 256, hipMemcpyHostToDevice);
		hipMemcpy(df, flags.data(), 256, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, din, dk, df, dout);
		std::vector<Out> out(64);
		if (hipMemcpy(out.data(), dout, 64 * sizeof(Out), hipMemcpyDeviceToHost) != hipSuccess) {
			printf("hip error\n");
			return 2;
		}
		for (int l = 0; l < 64; l++) {
			const int rb = l & 48, rl = l & 15;
			const uint32_t eshr = rl == 0 ? 0xAAAAu : in[l - 1], eshl = rl == 15 ? 0xBBBBu : in[l + 1];
			uint32_t emn = 0xFFFFFFFFu;
			for (int k = 0; k < 16; k++)
				emn = in[rb + k] < emn ? in[rb + k] : emn;
			const uint32_t erd = in[rb + ((rl * 7 + 3) & 15)];
			uint32_t edup = 0;
			for (int k = 0; k < rl; k++)
				if (flags[rb + k] && keys[rb + k] == keys[l])
					edup = 1;
			// argmin over (key<<32 | v&3, v): smallest key, then seq; lowest lane among equals
			int el = -1;
			unsigned long long ek = 0;
			uint32_t es = 0;
			for (int k = 0; k < 16; k++) {
				const unsigned long long kk = ((unsigned long long)keys[rb + k] << 32) | (in[rb + k] & 3u);
				const uint32_t ss = in[rb + k];
				if (el < 0 || kk < ek || (kk == ek && ss < es)) {
					el = k;
					ek = kk;
					es = ss;
				}
			}
			const Out& o = out[l];
			if (o.shr != eshr || o.shl != eshl || o.mn != emn || o.rd != erd || o.dup != edup || (int)o.argLane != el || o.argKey != ek || o.argSeq != es) {
				if (bad < 10)
					printf("trial %d lane %d: shr %x/%x shl %x/%x min %x/%x read %x/%x dup %u/%u arg %u/%d key %llx/%llx seq %x/%x\n", trial, l, o.shr, eshr, o.shl, eshl, o.mn,
						emn, o.rd, erd, o.dup, edup, o.argLane, el, o.argKey, ek, o.argSeq, es);
				bad++;
			}
		}
	}
	printf(bad ? "FAILED %d\n" : "row primitives OK\n", bad);
	return bad ? 1 : 0;
}

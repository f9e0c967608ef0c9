/* Host check of the division the device code uses for resolutions (pp_device.hpp: div_by):
 *   q0 = a * y;  e = fma(-q0, b, a);  q = fma(e, y, q0)      with y = 1.0 / b
 * must give the bits of a / b.  Operands: the resolutions the planners use, random numerators of map magnitude and the
 * +-4 ulp neighbourhoods of exact multiples of b (where a truncation after the division would flip).
 * Usage: test_reciprocal_division [iterations per divisor]; exit code 0 = identical everywhere. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double div_by(double a, double b, double y)
{
	const double q0 = a * y;
	const double e = fma(-q0, b, a);
	return fma(e, y, q0);
}

int main(int argc, char** argv)
{
	const long iters = argc > 1 ? atol(argv[1]) : 20000000L;
	const double bs[] = { (double)0.1f, 1.0, 0.0872664600610733, (double)0.05f, 0.1, 0.5, 1.5, (double)0.2f, 6.283185307179586 / 72, 0.0872664625997165 };
	uint64_t s = 88172645463325252ull;
	long bad = 0;
	for (unsigned bi = 0; bi < sizeof bs / sizeof bs[0]; bi++) {
		const double b = bs[bi], y = 1.0 / b;
		for (long it = 0; it < iters; it++) {
			s ^= s << 13;
			s ^= s >> 7;
			s ^= s << 17;
			double a;
			if (it & 1) {
				a = ((double)(int64_t)s / 9.2e18) * 1.0e4;
			} else {
				const long k = (long)((s >> 20) % 200000) - 100000;
				a = (double)k * b;
				const int d = (int)((s >> 5) % 9) - 4;
				uint64_t bits;
				memcpy(&bits, &a, 8);
				bits += (uint64_t)(int64_t)d;
				memcpy(&a, &bits, 8);
			}
			const double q = a / b, qm = div_by(a, b, y);
			if (memcmp(&q, &qm, 8) != 0 && !(q != q && qm != qm)) {
				if (bad < 5)
					printf("b=%.17g a=%.17g: a/b=%.17g div_by=%.17g\n", b, a, q, qm);
				bad++;
			}
		}
	}
	printf("%ld mismatches\n", bad);
	return bad ? 1 : 0;
}

"""How a persistent ray-cast wave takes its rays (struct WorkClaim, chroma_amd/csrc/kernel_step_control.h): part of every
wave's share owned by position, the rest through the work counter.  The struct's text is compiled for the host as it stands
(with stand-ins for the four device built-ins it uses) and driven by a simulation of the waves in random interleavings:
for every launch size, grid, claim size and share, every ray index must be handed out exactly once, nothing beyond the
last ray, and a launch of at most one chunk per wave must not touch the counter.  Host code only -- no GPU."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r'''
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
using std::min; using std::max;
#define __device__
struct Dim { unsigned x; };
static Dim gridDim, blockIdx;
static unsigned long long n_atomics;
static uint32_t atomicAdd(uint32_t *p, uint32_t v) { uint32_t o = *p; *p += v; n_atomics++; return o; }
static int __builtin_amdgcn_readfirstlane(int v) { return v; }
%s
struct Wave { WorkClaim wc; bool exhausted, done; unsigned id; };
static uint32_t rnd_state = 12345u;
static uint32_t rnd() { rnd_state = rnd_state * 1664525u + 1013904223u; return rnd_state >> 8; }
int main()
{
    const int big_chunk = 64;
    long long cases = 0;
    const unsigned grids[] = {1, 3, 64, 257, 6144};
    for (unsigned grid : grids) {
        std::vector<long long> sizes = {1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 1000, 4097};
        for (long long k : {1ll, 2ll, 3ll, 4ll, 5ll, 8ll, 9ll})
            for (long long d : {-17ll, -1ll, 0ll, 1ll, 16ll, 77ll}) {
                sizes.push_back(16ll * grid * k + d);
                sizes.push_back((long long)big_chunk * grid * k + d);
            }
        for (long long n : sizes) {
            if (n <= 0 || n > 40000000ll) continue;
            for (int big = 0; big <= 8; big++) for (int small = 0; small <= 8; small += (big == 5 ? 1 : 4)) {
                const int eighths = big | small << 4;
                const int chunk = (n > 4ll * big_chunk * (long long)grid) ? big_chunk : 16;
                gridDim.x = grid;
                uint32_t counter = 0;
                n_atomics = 0;
                std::vector<uint8_t> seen((size_t)n, 0);
                std::vector<Wave> waves;
                for (unsigned b = 0; b < grid; b++) {
                    if ((long long)b * 16 >= n) continue;              // (the kernels' early exit)
                    blockIdx.x = b;
                    waves.push_back(Wave{WorkClaim((int)n, chunk, eighths), false, false, b});
                }
                size_t live = waves.size();
                while (live) {
                    Wave &w = waves[rnd() %% waves.size()];
                    if (w.done) continue;
                    if (w.exhausted) { w.done = true; live--; continue; }
                    blockIdx.x = w.id;
                    const uint32_t base = w.wc.next(&counter, 0, w.exhausted);
                    const uint32_t lo = min(base, (uint32_t)n), hi = min(base + (uint32_t)chunk, (uint32_t)n);
                    for (uint32_t i = lo; i < hi; i++) {
                        if (seen[i]) { printf("ray %%u twice: n %%lld grid %%u chunk %%d eighths %%#x\n", i, n, grid, chunk, eighths); return 1; }
                        seen[i] = 1;
                    }
                }
                for (long long i = 0; i < n; i++)
                    if (!seen[(size_t)i]) { printf("ray %%lld never: n %%lld grid %%u chunk %%d eighths %%#x\n", i, n, grid, chunk, eighths); return 1; }
                const int e = chunk == 16 ? small : big;
                const long long nchunks = (n + chunk - 1) / chunk;
                if (e && nchunks <= (long long)waves.size() && n_atomics) { printf("counter touched: n %%lld grid %%u eighths %%#x\n", n, grid, eighths); return 1; }
                if (e == 8 && n_atomics) { printf("a full share must not touch the counter: n %%lld grid %%u\n", n, grid); return 1; }
                if (!e && n_atomics < (unsigned long long)nchunks) { printf("eighths 0 must claim every chunk through the counter\n"); return 1; }
                cases++;
            }
        }
    }
    printf("ok %%lld\n", cases);
    return 0;
}
'''


def test_every_ray_is_handed_out_exactly_once(tmp_path):
    src = open(os.path.join(ROOT, 'chroma_amd', 'csrc', 'kernel_step_control.h')).read()
    m = re.search(r'^struct WorkClaim \{.*?^\};', src, re.S | re.M)
    assert m, 'struct WorkClaim not found'
    cpp = tmp_path / 'work_claim_test.cpp'
    cpp.write_text(HARNESS % m.group(0))
    exe = tmp_path / 'work_claim_test'
    subprocess.run(['g++', '-O2', '-std=c++17', '-o', str(exe), str(cpp)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith('ok ') and int(out.stdout.split()[1]) > 3000, out.stdout

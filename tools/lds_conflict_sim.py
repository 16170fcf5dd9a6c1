#!/usr/bin/env python3
"""LDS bank conflicts of the oscillators' table lookups, simulated (no GPU): what layouts could change, and what they cannot.

Every oscillator sample reads the pair E[a], E[a + 1] of the half-table image (device_util.hpp Table<1>) with one ds_read2_b32: two dword
accesses, each served in two groups of 32 lanes over 32 banks of 4 bytes (MI355X_MICROARCH.md "LDS": bank = (address / 4) mod 32; an extra
distinct address on a busy bank costs a cycle).  Lane l of a wavefront owns samples 4 l .. 4 l + 3 of the chunk, so the 32 lanes of a group
look up an arithmetic progression of phases, folded at the table's middle.  For the sweeps of BASELINE configs[2], [4] and [3] this script
counts LDS cycles per conflict-free cycle for
  * the image as it is (blocks of 32 entries at a pitch of 33 words), a linear image, an XOR swizzle, a pitch of 65;
  * lanes owning INTERLEAVED samples (lane l: samples l, l + 64, l + 128, l + 192) instead of four consecutive ones.
Result (profiles/r04_lds_conflict_sim.txt): the pitched image sits at 2.7-3.0 cycles per cycle = 63-66 % of the LDS-active cycles in
conflicts — exactly what SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE measure on the hardware (63.8 % headline, 66.7 % cfg5, 66.3 % cfg4: profiles/
r03_summary.md) — and NONE of the alternatives moves it: 32 lanes gathering from effectively random places in 32 banks collide like 32 balls
in 32 bins (expected fullest bin ~3), whatever the layout.  What would halve the LDS time is the instruction, not the layout — one 8-byte
ds_read_b64 per pair (64 banks, two cycles instead of four) — but that needs every pair in one aligned slot, i.e. every entry stored twice: 192 KB
against 160 KB of LDS.  So the layout stays; the kernels are bound by their vector instructions anyway."""
import numpy as np

SR = 48000
M = SR // 2


def cycles(addr_words, nbanks=32, group=32):
    tot = ideal = 0
    for g0 in range(0, 64, group):
        for row in addr_words[:, g0:g0 + group]:
            u = np.unique(row)
            tot += np.bincount(u % nbanks, minlength=nbanks).max()
            ideal += 1
    return tot, ideal


LAYOUTS = {"pitch 33 (as built)": lambda k: k + (k >> 5), "linear": lambda k: k, "xor swizzle": lambda k: k ^ ((k >> 5) & 31), "pitch 65": lambda k: k + (k >> 6)}


def run(freqs, interleaved, word, chunks=3):
    tot = ideal = 0
    lane = np.arange(64)
    for f in freqs:
        for g in range(chunks):
            for c in range(4):
                t = g * 256 + (lane + 64 * c if interleaved else 4 * lane + c)
                i = np.floor((f * (t + 1)) % SR).astype(np.int64)
                a = np.abs(i - M)
                a2, i2 = cycles(np.stack([word(a), word(a) + 1]))
                tot += a2
                ideal += i2
    return tot / ideal


SWEEPS = {"configs[2] f = 10 k": [10.0 * k for k in range(1, 1025, 7)], "configs[4] f = 20 + k/8": [20 + k / 8 for k in range(0, 65536, 449)],
          "configs[3] f = 110 + k/64": [110 + k / 64 for k in range(0, 8192, 57)]}
for name, fr in SWEEPS.items():
    for interleaved in (False, True):
        for lname, word in LAYOUTS.items():
            r = run(fr, interleaved, word)
            print("%-26s %-28s %-20s %5.2f LDS cycles per conflict-free cycle = %2.0f %% of them conflicts" %
                  (name, "lanes own l, l+64, l+128, l+192" if interleaved else "lanes own 4l .. 4l+3 (as built)", lname, r, 100 * (r - 1) / r))

"""Bank-conflict census of the bf16-storage 3x3 kernel's LDS patch reads (csrc/conv_bf16s.hip).
Patch pixel q occupies 16-B slots 5q..5q+3 (pitch 80 B: 32 channels + one pad slot); a ds_read_b128 is served in four groups of 16 lanes
({0-3,12-15,20-27}, {4-11,16-19,28-31}, the same +32); a group is conflict-free when its 16 addresses fall in 16 different 16-B columns of
the 256-B bank row.  Lane li of a 32-pixel block = (row li / BW, column (li % BW - rot[row]) % BW)."""
import itertools, sys
GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
def worst(BW, PW, rot, pitch_slots=5):
    R = 32 // BW
    w = 1
    for r, s in itertools.product(range(3), range(3)):
        for grp in GROUPS:
            cols = {}
            for li in grp:
                row, j = li // BW, li % BW
                col = (j - rot[row]) % BW
                q = (row + r) * PW + col + s
                c = (pitch_slots * q) % 16
                cols[c] = cols.get(c, 0) + 1
            w = max(w, max(cols.values()))
    return w
for BW, PWs in ((32, (34,)), (16, (18,)), (8, (10,))):
    R = 32 // BW
    for PW in PWs:
        best = None
        for rot in itertools.product(range(BW), repeat=R - 1):
            rot = (0,) + rot
            wv = worst(BW, PW, rot)
            if best is None or wv < best[0]:
                best = (wv, rot)
            if wv == 1:
                break
        print('BW=%d PW=%d: worst %d-way with rotations %s' % (BW, PW, best[0], best[1]))

"""Generator of the inline-asm block `compute(s)` of wgrad_bf16_body (csrc/conv_bf16.hip): one step = 2 output rows x 32 pixels of a strip,
36 MFMAs (2 k-steps of 16 pixels x 2 output rows x 9 taps) per wave.

Operands of the block: %0..%8 accumulators of taps t = 3a + b; %9..%12 per-lane LDS addresses of input rows 2s + r (r = 0..3) of this
step; %13, %14 of dz rows yy = 0, 1.  A fragment A(r, b, ks) = input row r, column shift b, k-step ks: two transposing reads at offsets
64 b + 1024 ks and + 256; B(yy, ks) = dz row yy: offsets 1024 ks and + 256.

Round 3: input row r serves tap a = r of output row 0 AND tap a = r - 1 of output row 1, so A(r, b, ks) is read ONCE and used by both MFMAs
(rows 1 and 2): 24 A fragments per step instead of 36 -- 56 transposing reads instead of 80 for the same 36 MFMAs (the kernel is LDS-read
bound: 1 KB of fragment reads per MFMA and wave was the LDS's full rate at a busy matrix pipe).  Both dz rows of a k-step are live together.

Registers: v[80:111] fixed: A ring of 4 fragments (v80..v95), B fragments (yy, ks) -> v[96 + 4 (2 ks + yy) ...].  LDS operations retire in
order, so every wait is a counted lgkmcnt.

--dma: the step's five LDS-DMA instructions (the wave's share of staging pass s + 3) are placed INSIDE the MFMA stream, one every seven
MFMAs: issued back to back in front of the block they cost ~64 cycles each on the wave's issue path (5 x 64 of a 1152-cycle step: the
ablation showed 0.03-0.045 ms of every 0.15-0.18 ms launch was DMA time the MFMA stream did not cover); behind an MFMA one costs ~15.
Operands: %[a0]..%[a4] 64-bit per-lane source addresses, %[mb] the LDS byte address of the row slot (SGPR), %[half] = lanes 0..31 (the
fifth piece is half a wave), %[ex] scratch SGPR pair.
usage: python scripts/gen_wgrad_bf16_step.py [--dma]  (prints the C string lines)"""
import sys
A_RING = 4
PREFETCH = 3          # A fragments in flight ahead of the one being consumed


def main():
    dma = "--dma" in sys.argv
    frags = []        # (name, kind, base operand, offset, register quad) in first-use order
    uses = []         # (A fragment index in `frags`, [(acc, B fragment name)])
    a_count = 0
    for ks in (0, 1):
        bname = {yy: "B%d%d" % (yy, ks) for yy in (0, 1)}
        for r in range(4):
            for b in range(3):
                need = []
                if r <= 2:
                    need.append((3 * r + b, bname[0]))
                if r >= 1:
                    need.append((3 * (r - 1) + b, bname[1]))
                for _, bn in need:                       # a B fragment enters the load order right before its first use
                    if bn not in [f[0] for f in frags]:
                        yy = int(bn[1])
                        frags.append((bn, "B", 13 + yy, 1024 * ks, 96 + 4 * (2 * ks + yy)))
                reg = 80 + 4 * (a_count % A_RING)
                frags.append(("A%d%d%d" % (r, b, ks), "A", 9 + r, 64 * b + 1024 * ks, reg))
                uses.append((len(frags) - 1, need))
                a_count += 1
    lines = []
    issued = []       # fragment indices in issue order
    done_at = {}      # fragment index -> number of read instructions issued when its second read was issued

    def load(i):
        name, kind, op, off, reg = frags[i]
        opn = "%%[x%d]" % (op - 9) if op < 13 else "%%[z%d]" % (op - 13)
        lines.append("ds_read_b64_tr_b16 v[%d:%d], %s offset:%d" % (reg, reg + 1, opn, off))
        lines.append("ds_read_b64_tr_b16 v[%d:%d], %s offset:%d" % (reg + 2, reg + 3, opn, off + 256))
        issued.append(i)
        done_at[i] = 2 * len(issued)

    def wait_for(i):
        outstanding_after = 2 * len(issued) - done_at[i]
        assert outstanding_after <= 15
        lines.append("s_waitcnt lgkmcnt(%d)" % outstanding_after)

    index = {f[0]: i for i, f in enumerate(frags)}
    nxt = 0           # next fragment (in first-use order) not yet issued
    a_order = [u[0] for u in uses]
    # prologue: everything up to and including the PREFETCH-th A fragment
    stop = a_order[PREFETCH - 1]
    while nxt <= stop:
        load(nxt); nxt += 1
    waited = set()
    for k, (ai, need) in enumerate(uses):
        for acc, bn in need:
            for fi in (index[bn], ai):
                if fi not in waited:
                    wait_for(fi); waited.add(fi)
            # (all fragments issued before fi have landed too: in-order retirement)
            for fj in list(waited):
                pass
            areg, breg = frags[ai][4], frags[index[bn]][4]
            lines.append("v_mfma_f32_32x32x16_bf16 %%[c%d], v[%d:%d], v[%d:%d], %%[c%d]" % (acc, areg, areg + 3, breg, breg + 3, acc))
            n_m = sum(1 for l in lines if l.startswith("v_mfma"))
            if dma and (n_m - 3) % 7 == 0 and 0 <= (n_m - 3) // 7 < 5:
                kq = (n_m - 3) // 7
                if kq == 4:
                    lines.append("s_mov_b64 %[ex], exec")
                    lines.append("s_mov_b64 exec, %[half]")
                lines.append("s_add_i32 m0, %%[mb], 0x%x" % (0x400 * kq))
                lines.append("global_load_lds_dwordx4 %%[a%d], off" % kq)
                if kq == 4:
                    lines.append("s_mov_b64 exec, %[ex]")
        # this A fragment's buffer is free once its MFMAs are issued: bring in the next fragments up to PREFETCH A's ahead
        if k + PREFETCH < len(uses):
            stop = a_order[k + PREFETCH]
            while nxt <= stop:
                # a fragment may only overwrite an A buffer whose MFMAs were issued: ring distance guarantees it (PREFETCH < A_RING)
                load(nxt); nxt += 1
    n_reads = sum(1 for l in lines if l.startswith("ds_read"))
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    assert n_mfma == 36 and n_reads == 56 and nxt == len(frags), (n_mfma, n_reads)
    # drop waits that cannot matter (lgkmcnt(n) directly after a wait with a smaller or equal count and no read in between)
    out, last = [], None
    for l in lines:
        if l.startswith("s_waitcnt"):
            c = int(l[l.index("(") + 1:-1])
            if last is not None and last <= c:
                continue
            last = c
        elif l.startswith("ds_read"):
            last = None
        out.append(l)
    for l in out:
        print('            "%s\\n\\t"' % l)
    print("// %d reads, %d MFMAs%s" % (n_reads, n_mfma, ", 5 DMAs" if dma else ""))


if __name__ == "__main__":
    main()

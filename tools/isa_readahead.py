#!/usr/bin/env python3
"""How far ahead of its first use does the compiled kernel issue each LDS weight-fragment read?

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S nerf_shared_amd/csrc/mlp_bf16_s16.hip -o /tmp/s16.s
    python tools/isa_readahead.py /tmp/s16.s mlp_bf16_s16_kernelILi10ELi4ELb1E

For every ds_read_b128 of a kernel's straight-line body: the number of v_mfma instructions issued between the
read and the first MFMA that takes the destination registers as its A operand.  0 means the MFMA waits for the
full LDS latency of its own fragment (the software read-ahead queue of pipeline.h was undone by the
scheduler); the histogram is the thing to look at after touching the pipeline."""
import collections
import re
import sys


def kernel_body(path, tag):
    """Instruction lines of the first kernel whose mangled name contains `tag` (from its label to s_endpgm)."""
    out, on = [], False
    with open(path) as f:
        for line in f:
            if not on:
                head = line.split()[0] if line.split() else ""
                on = head.startswith("_ZN") and head.endswith(":") and tag in head
                continue
            out.append(line.strip())
            if line.strip().startswith("s_endpgm"):
                break
    return out


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main(path, tag):
    body = kernel_body(path, tag)
    pending = []          # (dst regs, mfma count at issue)
    n_mfma = 0
    hist = collections.Counter()
    waits0 = 0
    for ins in body:
        parts = ins.replace(",", " ").split()
        if not parts:
            continue
        op = parts[0]
        if op == "ds_read_b128":
            dst = regs(parts[1])
            pending = [p for p in pending if not (p[0] & dst)]
            pending.append((dst, n_mfma))
        elif op.startswith("v_mfma"):
            a = regs(parts[2])
            for p in list(pending):
                if p[0] & a:
                    hist[n_mfma - p[1]] += 1
                    pending.remove(p)
            n_mfma += 1
        elif op == "s_waitcnt" and "lgkmcnt(0)" in ins:
            waits0 += 1
    total = sum(hist.values())
    print("kernel %s: %d MFMAs, %d fragment reads matched, %d s_waitcnt lgkmcnt(0)" % (tag, n_mfma, total, waits0))
    for d in sorted(hist):
        print("  read issued %2d MFMAs before its first use: %5d (%.1f %%)" % (d, hist[d], 100.0 * hist[d] / max(total, 1)))
    return hist


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

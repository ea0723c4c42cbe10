"""Static check of the counted vmcnt waits of the fused MLP kernels against their compiled ISA.

The weight-ring syncs (csrc/pipeline.h block_sync) wait with `s_waitcnt vmcnt(N)` where N counts the
ring-DMA pieces *and* the compiler-issued global stores that are younger than the block being
published (the LEDGER).  An over-count would publish a block whose DMA has not landed.  This script
replays the straight-line instruction stream of a kernel (`hipcc -S` output), keeps the in-order
VMEM queue, and checks at every `s_waitcnt vmcnt(N); s_barrier` pair that all pieces of the block
the sync publishes are older than the N youngest operations.  It also reports how many younger DMA
pieces each wait leaves in flight (the point of the ledger).

usage: python tools/check_vmcnt.py file.s [kernel-name-substring]
"""
import re
import sys


def check(path, want="", verbose=True, slack=0, lag=0):
    """Replay every ring kernel in `path` whose symbol contains `want`.  `slack` is added to every
    sync's vmcnt (negative = pretend the ledger claimed more stores; used to test the checker).
    `lag`: the DMA issue phase of the waves to replay when the kernel selects its DMA sites inside the asm statement
    (pipeline.h SPLIT_DMA: `s_cmp_lg_u32 phase, K` + a branch over the site of phase K; two phases: 0 and 2); every
    phase must pass.  The ring DMA is global_load_lds or buffer_load ... lds."""
    text = open(path).read()
    funcs = re.split(r"\n(?=_Z[_A-Za-z0-9.$]+:)", text)       # function symbols only; local labels stay inside
    stats = {"ok": True, "kernels": 0, "syncs": 0, "dma_pieces": 0, "min_inflight": None}
    for f in funcs:
        name = f.split(":", 1)[0].strip()
        wants = (want,) if isinstance(want, str) else tuple(want)      # every substring must occur in the symbol
        if "s_barrier" not in f or not ("global_load_lds" in f or re.search(r"buffer_load_dwordx4 .* lds", f)) or not all(w in name for w in wants):
            continue
        stats["kernels"] += 1
        queue = []          # in issue order: ('dma', block) / ('st',) / ('ld',)
        n_dma = 0
        pieces = 2          # BF / WAVES
        n_sync = 0
        lines = f.splitlines()
        skip_to = None      # label that ends a DMA site this half does not execute
        pending_cmp = None
        for idx, line in enumerate(lines):
            stripped = line.strip()
            if skip_to is not None:
                if stripped.startswith(skip_to + ":"):
                    skip_to = None
                continue
            ins = stripped.split(" ")[0] if stripped else ""
            m_cmp = re.match(r"s_cmp_lg_u32 s\d+, (\d+)$", stripped)
            if m_cmp:                   # `s_cmp_lg_u32 phase, K; s_cbranch_scc1 .Lskip_...`: the site belongs to phase K
                pending_cmp = int(m_cmp.group(1))
                continue
            if ins == "s_cbranch_scc1" and pending_cmp is not None and ".Lskip_" in stripped:
                if lag != pending_cmp:
                    skip_to = stripped.split()[-1]
                pending_cmp = None
                continue
            pending_cmp = None
            if ins.startswith("global_load_lds") or (ins == "buffer_load_dwordx4" and stripped.endswith("lds")):
                queue.append(("dma", n_dma // pieces)); n_dma += 1
            elif ins.startswith(("global_store", "global_atomic", "scratch_store", "buffer_store", "flat_store")):
                queue.append(("st",))
            elif ins.startswith(("global_load", "scratch_load", "buffer_load", "flat_load")):
                queue.append(("ld",))
            elif ins == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", line)
                if m:       # everything but the N youngest has completed
                    n = int(m.group(1))
                    nxt = lines[idx + 1].strip().split(" ")[0] if idx + 1 < len(lines) else ""
                    if nxt == "s_barrier":
                        n = max(0, n - slack) if slack < 0 else n + slack
                    queue = queue[len(queue) - n:] if n else []
            elif ins == "s_barrier":
                if n_dma == 0:         # a barrier before any ring DMA (a tile loop's end-of-tile drain laid out in front
                    continue           # of the loop body): not one of the numbered syncs
                need = n_sync          # sync B = n_sync - 1 publishes block B + 1
                if any(q[0] == "dma" and q[1] <= need for q in queue):
                    stats["ok"] = False
                    if verbose:
                        print("  FAIL %s: sync %d leaves pieces of block %d in flight" % (name[:60], n_sync - 1, need))
                inflight = sum(1 for q in queue if q[0] == "dma")
                if n_sync > 0 and (stats["min_inflight"] is None or inflight < stats["min_inflight"]):
                    stats["min_inflight"] = inflight
                n_sync += 1
        stats["syncs"] += n_sync
        stats["dma_pieces"] += n_dma
        if verbose:
            print("%s\n  syncs %d, dma pieces %d, fewest younger DMA pieces a sync leaves in flight: %s"
                  % (name[:100], n_sync, n_dma, stats["min_inflight"]))
    return stats


if __name__ == "__main__":
    tag = sys.argv[2] if len(sys.argv) > 2 else ""
    sys.exit(0 if check(sys.argv[1], tag, lag=0)["ok"] and check(sys.argv[1], tag, lag=2)["ok"] else 1)

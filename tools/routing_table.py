#!/usr/bin/env python3
"""Layer -> kernel family for every 3x3x3 convolution of the BASELINE configurations (host logic only: runs without a GPU).

    python tools/routing_table.py > profiles/r04_routing_table.txt

One block per case of tmdiff_amd.routing.BASELINE_CASES (BASELINE.json configs[0..4]; B in {1, 8, 32}, 4 and 8 bands, 64^2 and
256^2 planes), one line per convolution of one inference forward (reference GeneralModel/Hyper_unet_general.py:600-636), then the
families each case reaches and -- last -- which families NO case reaches.  tests/test_host_logic.py asserts the same."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmdiff_amd import ops, routing  # noqa: E402


def main():
    reached, other = collections.Counter(), collections.Counter()
    print("# kernel family per 3x3x3 convolution (tmdiff_amd/routing.py; switches: defaults of ops.config)")
    print("# family -> C entry point / kernel: see the table at the top of tmdiff_amd/routing.py")
    for label, ch, b, n, size, math in routing.BASELINE_CASES + routing.OTHER_CASES:
        rows = routing.unet_table(ch, b, n, size, size, math)
        fams = collections.Counter(f for _, f in rows)
        (reached if (label, ch, b, n, size, math) in routing.BASELINE_CASES else other).update(fams)
        print(f"\n== {label}: channels {ch}, B = {b}, {n} bands, {size}x{size}, {math} -- "
              + ", ".join(f"{k} x{v}" for k, v in sorted(fams.items())))
        for L, fam in rows:
            split = ""
            if fam in ("wf", "wf_pair"):
                s = routing.wf_route(b, L.cin, L.cout, n, L.h, L.w, L.groups)[1]
                split = f" split-K {s}" if s > 1 else ""
            print(f"  {L.name:24s} {L.cin:4d}->{L.cout:<4d} g{L.groups} {n}x{L.h}x{L.w:<4d}"
                  f"{'' if L.plain else ' (segments)':11s} {fam}{split}")
    print("\n== families reached by the BASELINE cases: " + ", ".join(f"{k} ({v} layer instances)" for k, v in sorted(reached.items())))
    never = [f for f in routing.PRODUCT_FAMILIES + routing.FALLBACK_FAMILIES if f not in reached]
    print("== families reached by NO BASELINE case: " + (", ".join(never) if never else "none"))
    print("==   of these, reached by the reference's default / fixture widths (general-shape kernel, product path): "
          + (", ".join(f for f in never if f in other) or "none"))
    print("==   reached by neither (tmdiff_amd/fallback.py: even band counts other than 4 / 8, `fallback` test marker): "
          + (", ".join(f for f in never if f not in other) or "none"))


if __name__ == "__main__":
    main()

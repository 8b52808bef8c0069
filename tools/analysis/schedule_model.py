"""Offline list-scheduling model of the ambient-occlusion pass's claims on MEASURED tile costs (gpurun_out/tile_costs_*.npz,
tools/analysis/dump_tile_costs.py): what would an order / a claim size buy before anybody builds it?
Every XCD group's 256 workgroups claim from their group's list, then help the others in turn (the kernel's rule).
    python3 tools/analysis/schedule_model.py gpurun_out/tile_costs_bunny_1080p_ao.npz"""
import heapq
import sys

import numpy as np

WG = 256          # workgroups per group
OVERHEAD = 1200   # ticks (10 ns) a claim costs beyond its packets: barriers, the claim, the tile's records and frames (~12 us)
INFLATE = 1.2     # what a tile costs more when it is claimed out of spatial order (fitted: measured mixed / spatial passes)


def spatial_index(tile, tiles_x, rows, strip_tiles=2):
    x, row = tile % tiles_x, tile // tiles_x
    strip = x // strip_tiles
    return ((strip // 8) * rows + row) * strip_tiles + x % strip_tiles


def groups(d):
    tiles_x, rows = int(d["tiles_x"]), int(d["rows"])
    strips = (tiles_x + 1) // 2
    out, at = [], 0
    for g in range(8):
        size = ((strips + 7 - g) >> 3) * 2 * rows
        entries = d["order"][at:at + int(d["constants"][g][0])] & 0x03FFFFFF
        entries = np.array(sorted(entries, key=lambda t: spatial_index(int(t), tiles_x, rows)))
        out.append(d["costs"][entries].astype(np.float64))
        at += size
    return out


def simulate(queues):
    """queues: per group a list of claim durations (ticks) in claim order."""
    heads = [0] * 8
    free = [(0.0, g) for g in range(8) for _ in range(WG)]
    heapq.heapify(free)
    ends = []
    while free:
        t, home = heapq.heappop(free)
        for turn in range(8):
            g = (home + turn) % 8
            if heads[g] < len(queues[g]):
                c = queues[g][heads[g]]
                heads[g] += 1
                heapq.heappush(free, (t + c, home))
                break
        else:
            ends.append(t)
    return max(ends) / 1e5, sum(ends) / len(ends) / 1e5


def policy(c, kind, heavy=2.0, runway=2.0, parts=2, tail_parts=4):
    """Claim durations of one group's tiles (c: costs in spatial order)."""
    n = len(c)
    ref = np.sort(c)[n - 1 - (n - 1) // 4]
    if kind == "spatial":
        return list(c)
    is_heavy = c > heavy * ref
    hv = np.sort(c[is_heavy])[::-1]
    rest = c[~is_heavy]
    left = rest.sum() - np.concatenate(([0.0], np.cumsum(rest)[:-1]))
    budget = runway * ref * WG
    in_bulk = (left > budget) & (rest >= 0.25 * ref)
    bulk, rw = rest[in_bulk], rest[~in_bulk]
    if kind == "mixed":  # the rule in use: the runway by falling cost, out of spatial order
        costly = rw >= 0.25 * ref
        return list(hv) + list(bulk) + list(np.sort(np.where(costly, rw * INFLATE, rw))[::-1])
    if kind == "mixed_free":  # ... if the runway's tiles cost what they cost in spatial order (the model of round 5's notes)
        return list(hv) + list(bulk) + list(np.sort(rw)[::-1])
    if kind == "parts":  # heavy tiles in halves; bulk whole; the runway's costly tiles IN SPATIAL ORDER in `parts` pieces; cheap ones last, falling
        out = []
        for x in hv:
            out += [x / 2 + OVERHEAD / 2] * 2
        out += list(bulk)
        costly, cheap = rw[rw >= 0.25 * ref], rw[rw < 0.25 * ref]
        k = len(costly)
        for i, x in enumerate(costly):
            p = parts if i < k * 2 // 3 else tail_parts  # (the last third finer still)
            out += [x / p + OVERHEAD * (p - 1) / p] * p
        out += list(np.sort(cheap)[::-1])
        return out
    raise ValueError(kind)


def main():
    for path in sys.argv[1:]:
        d = np.load(path)
        gs = groups(d)
        total = sum(g.sum() for g in gs)
        print(f"{path}: {sum(len(g) for g in gs)} tiles, {total / 1e5:.1f} ms of workgroup time, ideal pass {total / 1e5 / (8 * WG):.4f} ms")
        for label, args in [("spatial, whole tiles", ("spatial",)), ("mixed (in use), runway tiles cost x%.2f" % INFLATE, ("mixed",)),
                            ("mixed, if the runway cost nothing extra", ("mixed_free",)),
                            ("spatial runway in halves, last third in quarters", ("parts", 2.0, 2.0, 2, 4)),
                            ("spatial runway in halves", ("parts", 2.0, 2.0, 2, 2)),
                            ("spatial runway in quarters", ("parts", 2.0, 2.0, 4, 4)),
                            ("spatial runway (3 claims deep) in halves / quarters", ("parts", 2.0, 3.0, 2, 4)),
                            ("spatial runway (1.5 deep) in halves / quarters", ("parts", 2.0, 1.5, 2, 4))]:
            last, mean = simulate([policy(g, *args) for g in gs])
            print(f"   {label:62s} last workgroup ends {last:.4f} ms, mean {mean:.4f}")


if __name__ == "__main__":
    main()

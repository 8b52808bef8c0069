"""Which order should the ambient-occlusion pass claim a frame's tiles in?  (GPU box.)
Measures the tiles' costs (rt_debug_measure_tile_costs), makes claim orders by several rules HERE, installs each
(rt_debug_set_tile_order: any order renders the same image) and times blocking frames with it.

    python3 tools/analysis/order_policies.py [workload ...]
"""
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def spatial_index(tile, tiles_x, rows, strip_tiles):
    x, row = tile % tiles_x, tile // tiles_x
    strip = x // strip_tiles
    return ((strip // 8) * rows + row) * strip_tiles + x % strip_tiles


def segments(order, constants, tiles_x, rows, strip_tiles):
    strips = (tiles_x + strip_tiles - 1) // strip_tiles
    out, at = [], 0
    for g in range(8):
        size = ((strips + 7 - g) >> 3) * strip_tiles * rows
        out.append((at, size, order[at:at + constants[g][0]].copy()))
        at += size
    return out


def make_order(base, segs, costs, rule, workgroups_per_group, tiles_x, rows, strip_tiles):
    order = base.copy()
    for at, size, entries in segs:
        tiles = entries & 0x03FFFFFF
        spatial = np.argsort([spatial_index(int(t), tiles_x, rows, strip_tiles) for t in tiles], kind="stable")
        entries, tiles = entries[spatial], tiles[spatial]
        c = costs[tiles].astype(np.float64)
        n = len(c)
        if n == 0:
            continue
        kind = rule[0]
        if kind == "spatial":
            pick = np.arange(n)
        elif kind == "lpt":
            pick = np.argsort(-c, kind="stable")
        elif kind == "buckets":  # ("buckets", heavy, runway, per_octave): like "mixed", but the runway falls by cost BUCKET
            # (`per_octave` buckets per factor of two), spatial order inside a bucket
            _, heavy, runway, per_octave = rule
            reference = np.sort(c)[n - 1 - (n - 1) // 4]
            is_heavy = c > heavy * reference
            heavy_ones = np.flatnonzero(is_heavy)
            heavy_ones = heavy_ones[np.argsort(-c[heavy_ones], kind="stable")]
            rest = np.flatnonzero(~is_heavy)
            left = c[rest].sum() - np.concatenate(([0.0], np.cumsum(c[rest])[:-1]))
            budget = runway * reference * workgroups_per_group
            in_spatial = (left > budget) & (c[rest] >= 0.25 * reference)
            sp, rw = rest[in_spatial], np.sort(rest[~in_spatial])
            bucket = np.floor(np.log2(np.maximum(c[rw], 1.0) / reference) * per_octave)
            rw = rw[np.argsort(-bucket, kind="stable")]
            pick = np.concatenate((heavy_ones, sp, rw))
        elif kind == "blocks":  # ("blocks", heavy, runway, size): like "mixed", but the runway falls by the mean cost of BLOCKS of `size` spatial neighbours
            _, heavy, runway, size = rule
            reference = np.sort(c)[n - 1 - (n - 1) // 4]
            is_heavy = c > heavy * reference
            heavy_ones = np.flatnonzero(is_heavy)
            heavy_ones = heavy_ones[np.argsort(-c[heavy_ones], kind="stable")]
            rest = np.flatnonzero(~is_heavy)
            left = c[rest].sum() - np.concatenate(([0.0], np.cumsum(c[rest])[:-1]))
            budget = runway * reference * workgroups_per_group
            in_spatial = left > budget
            sp, rw = rest[in_spatial], rest[~in_spatial]
            # cheap tiles of the spatial part join the runway, in place (spatial order is the index order)
            cheap = sp[c[sp] < 0.25 * reference]
            sp = sp[c[sp] >= 0.25 * reference]
            rw = np.sort(np.concatenate((rw, cheap)))
            blocks = [rw[i:i + size] for i in range(0, len(rw), size)]
            blocks.sort(key=lambda b: -float(c[b].mean()))
            pick = np.concatenate([heavy_ones, sp] + blocks) if blocks else np.concatenate((heavy_ones, sp))
        else:  # ("mixed", heavy, runway, cheap)
            _, heavy, runway, cheap = rule
            reference = np.sort(c)[n - 1 - (n - 1) // 4]
            is_heavy = c > heavy * reference
            heavy_ones = np.flatnonzero(is_heavy)
            heavy_ones = heavy_ones[np.argsort(-c[heavy_ones], kind="stable")]
            rest = np.flatnonzero(~is_heavy)
            left = c[rest].sum() - np.concatenate(([0.0], np.cumsum(c[rest])[:-1]))
            budget = runway * reference * workgroups_per_group
            in_spatial = (left > budget) & (c[rest] >= cheap * reference)
            sp, rw = rest[in_spatial], rest[~in_spatial]
            rw = rw[np.argsort(-c[rw], kind="stable")]
            pick = np.concatenate((heavy_ones, sp, rw))
        order[at:at + n] = entries[pick]
    return order


def simulate(order, segs, costs, workgroups_per_group=256):
    """List scheduling by the kernel's rules: every group's workgroups claim from their own queue, then help the others in
    turn.  Returns (makespan, mean end, first end) in ms of the measured costs (10 ns ticks)."""
    import heapq

    queues = [list(costs[order[at:at + len(e)] & 0x03FFFFFF]) for at, _, e in segs]
    heads = [0] * 8
    free = [(0.0, g, w) for g in range(8) for w in range(workgroups_per_group)]
    heapq.heapify(free)
    ends = []
    while free:
        t, home, w = heapq.heappop(free)
        for turn in range(8):
            g = (home + turn) % 8
            if heads[g] < len(queues[g]):
                c = queues[g][heads[g]]
                heads[g] += 1
                heapq.heappush(free, (t + c, home, w))
                break
        else:
            ends.append(t)
    return max(ends) / 1e5, sum(ends) / len(ends) / 1e5, min(ends) / 1e5


def timed(host, frames=30):
    for _ in range(5):
        host.render()
    total, ao = [], []
    for _ in range(frames):
        host.render()
        total.append(host.last_kernel_ms)
        ao.append(host.last_ao_ms)
    return statistics.median(total), statistics.median(ao), min(total)


def main():
    import opencl_raytracer_amd as rt
    from bench import WORKLOADS, load_scene, workload_options

    for name in sys.argv[1:] or ["bunny_1080p_ao"]:
        w = WORKLOADS[name]
        opt = workload_options(rt, w)
        scene = load_scene(rt, w).build_bvh(opt.bvh_method)
        host = rt.Host(opt, 0)
        host.upload_scene(scene)
        base_ms = timed(host)
        host.measure_tile_costs(3, reorder=False)
        info = host.tile_order()
        costs, words = info["costs"] / 3.0, info["words"]
        n = int(np.sqrt(opt.n_super_samples))
        tiles_x, rows = (opt.width * n + 7) // 8, (opt.height * n + 7) // 8
        strip_tiles = 2
        hit = words >> 8 != 0
        c = costs[hit]
        q = np.percentile(c, [5, 25, 50, 75, 90, 95, 99, 100])
        print(f"{name}: {hit.sum()} tiles with AO work; measured cost per tile (us, 10 ns ticks / 100): "
              f"5/25/50/75/90/95/99/100 % = " + " ".join(f"{v / 100:.1f}" for v in q) + f"; sum {c.sum() / 1e5:.1f} ms of workgroup time")
        cls = (words[hit] >> 8).astype(np.float64)
        print(f"   correlation of the measured cost with the cost class (primary packet's leaf stops): {np.corrcoef(cls, c)[0, 1]:.3f}")
        segs = segments(info["order"], info["constants"], tiles_x, rows, strip_tiles)
        sums = [float(costs[e & 0x03FFFFFF].sum()) / 1e5 for _, _, e in segs]
        print("   workgroup time per XCD group's queue (ms): " + " ".join(f"{v:.1f}" for v in sums) +
              f"; ideal pass with 256 workgroups each: {max(sums) / 256:.4f} ms (largest group), {sum(sums) / 2048:.4f} ms (all groups level)")
        print(f"   {'as installed (blocks of 64 by cost class)':58s} frame {base_ms[0]:.4f} ms (min {base_ms[2]:.4f}), ao {base_ms[1]:.4f}", flush=True)
        rules = [("spatial",), ("lpt",), ("mixed", 2.0, 2.0, 0.25), ("mixed", 1.6, 2.0, 0.25)]
        if os.environ.get("OCRT_ORDER_BLOCKS"):
            for size in (4, 8, 16, 32, 64):
                for runway in (2.0, 3.0, 6.0):
                    rules.append(("blocks", 2.0, runway, size))
        if os.environ.get("OCRT_ORDER_BUCKETS"):
            for per_octave in (1, 2, 4):
                for runway in (2.0, 3.0, 100.0):
                    rules.append(("buckets", 2.0, runway, per_octave))
        rules += [("mixed", 2.0, 2.0, 0.25)]
        for rule in rules:
            order = make_order(info["order"], segs, costs, rule, 256, tiles_x, rows, strip_tiles)
            host.set_tile_order(order, info["constants"])
            ms = timed(host)
            sim = simulate(order, segs, costs)
            print(f"   {str(rule):58s} frame {ms[0]:.4f} ms (min {ms[2]:.4f}), ao {ms[1]:.4f}   model: last workgroup ends {sim[0]:.4f}, mean {sim[1]:.4f}, first {sim[2]:.4f}", flush=True)
        # iterate: the costs measured UNDER the mixed order (the runway's tiles cost more when their neighbours are not being
        # walked beside them), the mixed order made again from those, measured again ...
        if os.environ.get("OCRT_ORDER_ITERATE"):
            current = costs.copy()
            for it in range(4):
                order = make_order(info["order"], segs, current, ("mixed", 2.0, 2.0, 0.25), 256, tiles_x, rows, strip_tiles)
                host.set_tile_order(order, info["constants"])
                ms = timed(host)
                sim = simulate(order, segs, current)
                print(f"   iteration {it}: mixed(2, 2) from the costs of iteration {it - 1 if it else 'spatial'}: frame {ms[0]:.4f} ms (min {ms[2]:.4f}), ao {ms[1]:.4f}"
                      f"   model: last {sim[0]:.4f}, mean {sim[1]:.4f}", flush=True)
                host.measure_tile_costs(3, reorder=False)
                current = host.tile_order()["costs"] / 3.0
                print(f"      costs under that order: sum {current[hit].sum() / 1e5:.1f} ms of workgroup time (under the first order {costs[hit].sum() / 1e5:.1f})", flush=True)
        # the library's own rule (DeviceRenderer::orderByMeasuredCost), with the heaviest tiles claimed half a tile at a time
        if os.environ.get("OCRT_ORDER_SPLIT"):
            host.measure_tile_costs(3, reorder=True)
            for split_above in (0.0, 1.0, 0.5, 0.35, 0.25, 0.18, 0.12, 0.08, 0.0):
                host.set_order_policy(2.0, 2.0, split_above)
                ms = timed(host)
                print(f"   library rule, tiles above {split_above:4.2f} of the pass's ideal length in halves: frame {ms[0]:.4f} ms (min {ms[2]:.4f}), ao {ms[1]:.4f}", flush=True)
        host.set_tile_order(info["order"], info["constants"])
        again = timed(host)
        print(f"   {'as installed, again':58s} frame {again[0]:.4f} ms (min {again[2]:.4f}), ao {again[1]:.4f}", flush=True)
        host.close()


if __name__ == "__main__":
    main()

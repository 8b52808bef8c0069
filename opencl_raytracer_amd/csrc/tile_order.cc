// tile_order.cc -- the claim orders of a frame's tiles, made once per upload on the host (DeviceRenderer members): the order
// the ambient-occlusion pass claims the tiles in (by cost class, or by measured cost), the order the primary pass takes its
// 2 x 2 blocks in, and how a stream of frames has its tiles' costs measured.
#include <algorithm>
#include <array>

#include "device_internal.h"

namespace ocrt {

// The order in which the ambient-occlusion pass claims the tiles (kernels/ao.hip.h): per XCD group -- the image's strips
// are dealt round-robin to eight groups, kernels/primary.hip.h -- a list of the group's non-empty tiles, entry = tile |
// (hit count - 1) << 26, costly tiles first so that the pass ends on short claims.  Camera, scene and options are fixed
// per upload, so the list is too: it is made here, once, on the host (until round 4 the last workgroup of every frame's
// primary pass sorted its group's tiles again).
// The rule (`tile_cost` empty): blocks of 64 neighbouring tiles -- a strip wide, 64 / strip_tiles high -- by the sum of
// their tiles' cost classes >> cost_shift, capped: every block of the model shares the top key and those blocks keep
// their spatial order (neighbouring claims walk the same part of the tree: scalar cache and L2 see it again; a finer
// key cost 3-10 % per frame, scene_pack.cc), the cheap blocks follow by cost; inside a block the tiles stay in spatial
// order.  With measured costs per tile (`tile_cost`, DeviceRenderer::measureTileCosts): see orderByMeasuredCost.
void DeviceRenderer::orderTiles() {
	const uint32_t strip_tiles = kp.strip_tiles, rows = kp.local_tile_rows, tiles_x = kp.tiles_x;
	const uint32_t strips = (tiles_x + strip_tiles - 1u) / strip_tiles;
	const size_t order_slots = (size_t) ((tiles_x + MAX_STRIP_TILES - 1) / MAX_STRIP_TILES) * MAX_STRIP_TILES * rows;
	std::vector<uint32_t> order(order_slots ? order_slots : 1, 0u);
	std::array<std::array<uint32_t, 3>, XCD_GROUPS> constants{};
	std::array<uint32_t, XCD_GROUPS> splits{};
	const bool measured = tile_cost.size() == tile_count;
	size_t segment = 0;
	for (uint32_t group = 0; group < XCD_GROUPS; ++group) {
		const uint32_t strips_here = (strips + XCD_GROUPS - 1u - group) >> 3;
		const uint32_t per_strip = strip_tiles * rows;
		const uint32_t tiles_here = strips_here * per_strip;  // incl. possible columns past the image
		// element e of the group: strip e / per_strip, then row-major across the strip
		struct Element {
			uint32_t tile, word;
		};
		std::vector<Element> elements;
		elements.reserve(tiles_here);
		for (uint32_t e = 0; e < tiles_here; ++e) {
			const uint32_t strip_index = e / per_strip, within = e - strip_index * per_strip;
			const uint32_t local_row = within / strip_tiles;
			const uint32_t tile_x = strip_tiles * (group + XCD_GROUPS * strip_index) + (within - local_row * strip_tiles);
			const uint32_t tile = local_row * tiles_x + tile_x;
			elements.push_back(Element{ tile, tile_x < tiles_x ? tile_words[tile] : 0u });
		}
		uint32_t work = 0, cost_total = 0, hit_total = 0;
		for (const Element &el : elements) {
			hit_total += el.word & 0xFFu;
			if (el.word >> 8) {
				++work;
				cost_total += el.word >> 8;
			}
		}
		std::vector<uint32_t> listed;  // indices into `elements`, in claim order
		listed.reserve(work);
		if (measured) {
			std::vector<uint32_t> candidates;
			for (uint32_t e = 0; e < tiles_here; ++e)
				if (elements[e].word >> 8)
					candidates.push_back(e);
			std::vector<float> cost(candidates.size());
			for (size_t i = 0; i < candidates.size(); ++i)
				cost[i] = tile_cost[elements[candidates[i]].tile];
			uint32_t split = 0;
			for (uint32_t at : orderByMeasuredCost(cost, &split))
				listed.push_back(candidates[at]);
			splits[group] = split;
		} else {
			const uint32_t n_blocks = (tiles_here + 63u) >> 6;
			std::vector<std::pair<uint32_t, uint32_t>> blocks;  // (key, block), stable by block
			for (uint32_t block = 0; block < n_blocks; ++block) {
				uint32_t cost = 0;
				for (uint32_t e = block * 64u; e < tiles_here && e < block * 64u + 64u; ++e)
					cost += elements[e].word >> 8;
				uint32_t key = kp.debug_no_sort ? 1u : 1u + (cost >> kp.cost_shift);
				blocks.push_back({ key > 64u ? 64u : key, block });
			}
			std::stable_sort(blocks.begin(), blocks.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
			for (const auto &b : blocks)
				for (uint32_t e = b.second * 64u; e < tiles_here && e < b.second * 64u + 64u; ++e)
					if (elements[e].word >> 8)
						listed.push_back(e);
		}
		for (size_t i = 0; i < listed.size(); ++i) {
			const Element &el = elements[listed[i]];
			order[segment + i] = el.tile | (((el.word & 0xFFu) - 1u) << 26);  // (tile: 26 bits, at most 2^32 sub-pixels per frame)
		}
		constants[group] = { work, cost_total, hit_total };
		segment += tiles_here;
	}
	installOrder(order, constants, splits);
}

// Claim order of one group's tiles from their MEASURED costs (device-clock ticks a tile's claims kept their workgroups
// busy, measureTileCosts): `cost` in the tiles' spatial order, returns the indices in claim order.
// What matters is how the pass ENDS: a claim is a whole tile for a workgroup's four waves, a costly tile keeps them
// busy ~0.15 ms of a 1 ms pass and some tiles cost four times the median -- claimed in spatial order, the last third of
// the pass ran at falling occupancy (profiles/r05_notes.md).  So:
//   1. tiles whose cost stands out (beyond `heavy` x the reference cost = the upper quartile) go first, costliest first;
//   2. the others follow IN SPATIAL ORDER (neighbouring claims walk the same part of the tree) ...
//   3. ... up to the RUNWAY: what would keep each workgroup of the group busy for about `runway` reference claims is held
//      back and claimed last by falling cost, so that the pass ends on its cheapest tiles; tiles below a quarter of the
//      reference cost anywhere in the list are moved there too.
//   4. `*split` (out): how many tiles at the head of the list -- heavy ones -- are to be claimed HALF A TILE at a time
//      (kernels/ao.hip.h): those whose cost exceeds `split_above` x the pass's ideal length (the group's total cost over its
//      workgroups).  A tile that keeps a workgroup for a quarter of the pass decides when the pass ends -- at 600 x 600 -s 4
//      the costliest tile takes as long as the whole pass should, and so it does in one GPU's share of a frame split eight ways.
std::vector<uint32_t> DeviceRenderer::orderByMeasuredCost(const std::vector<float> &cost, uint32_t *split) const {
	const size_t n = cost.size();
	std::vector<uint32_t> out;
	out.reserve(n);
	if (split)
		*split = 0;
	if (n == 0)
		return out;
	std::vector<float> sorted(cost);
	std::sort(sorted.begin(), sorted.end());
	const float reference = sorted[n - 1 - (n - 1) / 4];
	const auto falling = [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; };
	std::vector<uint32_t> heavy, rest;
	double heavy_from = (double) order_policy.heavy * reference;
	if (split && order_policy.split_above > 0.0f) {  // (whatever is to be split must stand in the heavy prefix)
		double total = 0.0;
		for (float c : cost)
			total += c;
		const uint32_t wg = aoWorkgroups() / XCD_GROUPS ? aoWorkgroups() / XCD_GROUPS : 1u;
		heavy_from = std::min(heavy_from, (double) order_policy.split_above * total / wg);
	}
	for (uint32_t i = 0; i < n; ++i)
		(cost[i] > heavy_from ? heavy : rest).push_back(i);
	std::stable_sort(heavy.begin(), heavy.end(), falling);
	const uint32_t workgroups = aoWorkgroups() / XCD_GROUPS ? aoWorkgroups() / XCD_GROUPS : 1u;
	if (split && order_policy.split_above > 0.0f && kp.shared_walk && (kp.ao_dirs & 1u) == 0u && kp.ao_dirs >= 2u) {
		double total = 0.0;
		for (float c : cost)
			total += c;
		const double ideal = total / workgroups;  // what the pass takes when the group's workgroups share its work evenly
		for (uint32_t i : heavy)  // (by falling cost: a prefix)
			if (cost[i] > order_policy.split_above * ideal)
				++*split;
	}
	const double budget = (double) order_policy.runway * reference * workgroups;
	double left = 0.0;
	for (uint32_t i : rest)
		left += cost[i];
	std::vector<uint32_t> spatial, runway;
	for (uint32_t i : rest) {
		(left > budget && cost[i] >= 0.25f * reference ? spatial : runway).push_back(i);
		left -= cost[i];
	}
	std::stable_sort(runway.begin(), runway.end(), falling);
	out.insert(out.end(), heavy.begin(), heavy.end());
	out.insert(out.end(), spatial.begin(), spatial.end());
	out.insert(out.end(), runway.begin(), runway.end());
	return out;
}

// The fused frame kernel's primary work (kernels/primary.hip.h, primary_top_up): per XCD group the 2 x 2 tile blocks of
// its strips, entry = first tile column | first tile row << 16, IN THE ORDER THE AMBIENT-OCCLUSION CLAIMS WANT THEM --
// a block is listed when the first of its tiles comes up in the group's claim order (`order_host`) -- and the blocks no
// claim ever wants (the background, tiles without hits) last, in spatial order.  order_need[j], beside entry j of the claim
// order: how many blocks the entries 0 ... j need, i.e. how far the block cursor must have come before entry j's tile may
// be waited for.
void DeviceRenderer::orderPrimaryBlocks() {
	const uint32_t strip_tiles = kp.strip_tiles, columns = strip_tiles >> 1, rows = kp.local_tile_rows, tiles_x = kp.tiles_x;
	const uint32_t strips = (tiles_x + strip_tiles - 1u) / strip_tiles, row_blocks = (rows + 1u) >> 1;
	const uint32_t blocks_x = (tiles_x + 1u) >> 1;
	primary_order_host.clear();
	order_need_host.assign(order_host.size(), 0u);
	std::vector<char> listed((size_t) blocks_x * row_blocks, 0);
	size_t ao_segment = 0;
	for (uint32_t group = 0; group < XCD_GROUPS; ++group) {
		const uint32_t strips_here = (strips + XCD_GROUPS - 1u - group) >> 3;
		const size_t segment = (size_t) strips_here * row_blocks * columns, at = primary_order_host.size();
		primary_order_host.resize(at + segment, 0u);
		uint32_t count = 0;
		// in the order of the claims
		for (uint32_t j = 0; j < queue_static[group][0]; ++j) {
			const uint32_t tile = order_host[ao_segment + j] & 0x03FFFFFFu, x = tile % tiles_x, row = tile / tiles_x;
			const size_t block = (size_t) (row >> 1) * blocks_x + (x >> 1);
			if (!listed[block]) {
				listed[block] = 1;
				primary_order_host[at + count++] = (x & ~1u) | (row & ~1u) << 16;
			}
			order_need_host[ao_segment + j] = count;
		}
		// ... then whatever no claim wants
		for (uint32_t strip_index = 0; strip_index < strips_here; ++strip_index)
			for (uint32_t rb = 0; rb < row_blocks; ++rb)
				for (uint32_t c = 0; c < columns; ++c) {
					const uint32_t x0 = strip_tiles * (group + XCD_GROUPS * strip_index) + 2u * c, row0 = 2u * rb;
					if (x0 >= tiles_x)
						continue;
					const size_t block = (size_t) rb * blocks_x + (x0 >> 1);
					if (!listed[block]) {
						listed[block] = 1;
						primary_order_host[at + count++] = x0 | row0 << 16;
					}
				}
		primary_blocks[group] = count;
		ao_segment += (size_t) strips_here * strip_tiles * rows;
	}
	if (primary_order_host.empty())
		primary_order_host.push_back(0u);
	orderBlocksByCost();
}

// primary_kernel's list: workgroup `seq` of a group takes entry `seq` of the group's part -- its 2 x 2 blocks of tiles by
// falling cost (the largest cost class among a block's tiles = the leaves its primary packet stops at; a tile without
// ambient-occlusion work carries none: its hit count stands in), spatial order among equals, the background last.  A tile
// of cost class `primary_split_above` or more gets an entry of its own, in front: the four waves of a workgroup cast a
// QUARTER of it each (kernels/primary.hip.h, primary_tile) -- such a tile's one wave would be the critical path of the pass.
// Entry: x | row << 13 | (waves with nothing to do) << 26 | (quarters) << 30; `kp.primary_list_stride` entries per group.
void DeviceRenderer::orderBlocksByCost() {
	const uint32_t strip_tiles = kp.strip_tiles, columns = strip_tiles >> 1, rows = kp.local_tile_rows, tiles_x = kp.tiles_x;
	const uint32_t strips = (tiles_x + strip_tiles - 1u) / strip_tiles, row_blocks = (rows + 1u) >> 1;
	blocks_by_cost_host.clear();
	kp.primary_list_stride = 0;
	if (!PRIMARY_BY_COST_OK(kp) || tile_words.size() != (size_t) tiles_x * rows)
		return;
	std::array<std::vector<std::pair<uint32_t, uint32_t>>, XCD_GROUPS> lists;  // (cost, entry)
	size_t stride = 0;
	// Quarters shorten the pass where a FEW tiles are its critical path; they are extra work (a quarter's packet walks
	// most of what the tile's did) where the chip's wave slots are full anyway: a 2 M-triangle height field, nearly every
	// tile of which stops at 64 leaves or more, casts its primary rays in 0.31 ms whole and 0.42 ms in quarters.  So: only
	// if the tiles that qualify, four waves each, take no more than `primary_split_waves` of the 8 192 wave slots the chip
	// has at one time (256 CUs x 32 waves).
	uint32_t split_above = order_policy.primary_split_above;
	if (split_above) {
		size_t qualifying = 0;
		for (uint32_t word : tile_words)
			qualifying += (word >> 8) >= split_above;
		primary_quartered = qualifying;
		if (qualifying * 4u > order_policy.primary_split_waves) {
			split_above = 0;
			primary_quartered = 0;
		}
	} else {
		primary_quartered = 0;
	}
	for (uint32_t group = 0; group < XCD_GROUPS; ++group) {
		const uint32_t strips_here = (strips + XCD_GROUPS - 1u - group) >> 3;
		auto &list = lists[group];
		for (uint32_t strip_index = 0; strip_index < strips_here; ++strip_index)
			for (uint32_t rb = 0; rb < row_blocks; ++rb)
				for (uint32_t c = 0; c < columns; ++c) {
					const uint32_t x0 = strip_tiles * (group + XCD_GROUPS * strip_index) + 2u * c, row0 = 2u * rb;
					if (x0 >= tiles_x)
						continue;
					uint32_t cost = 0, idle = 0;
					for (uint32_t k = 0; k < 4u; ++k) {
						const uint32_t x = x0 + (k & 1u), row = row0 + (k >> 1);
						if (x >= tiles_x || row >= rows) {
							idle |= 1u << k;
							continue;
						}
						const uint32_t word = tile_words[(size_t) row * tiles_x + x];
						const uint32_t tile_cost_class = word >> 8 ? word >> 8 : (word & 0xFFu) ? 1u : 0u;
						if (split_above && tile_cost_class >= split_above) {
							list.push_back({ tile_cost_class, x | row << 13 | 1u << 30 });
							idle |= 1u << k;
						} else {
							cost = std::max(cost, tile_cost_class);
						}
					}
					if (idle != 15u)
						list.push_back({ cost, x0 | row0 << 13 | idle << 26 });
				}
		std::stable_sort(list.begin(), list.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
		stride = std::max(stride, list.size());
	}
	if (stride == 0)
		return;
	blocks_by_cost_host.assign(XCD_GROUPS * stride, 0xFFFFFFFFu);
	for (uint32_t group = 0; group < XCD_GROUPS; ++group)
		for (size_t i = 0; i < lists[group].size(); ++i)
			blocks_by_cost_host[group * stride + i] = lists[group][i].second;
	kp.primary_list_stride = (uint32_t) stride;
}

// ... and onto the device (the list grows with the tiles cast in quarters)
void DeviceRenderer::uploadBlocksByCost() {
	if (blocks_by_cost_host.empty())
		return;
	if (blocks_by_cost_host.size() > blocks_by_cost_capacity) {
		device_free(d_blocks_by_cost);
		d_blocks_by_cost = nullptr;
		blocks_by_cost_capacity = blocks_by_cost_host.size();
		d_blocks_by_cost = device_alloc(blocks_by_cost_capacity * sizeof(uint32_t));
	}
	OCRT_HIP(hipMemcpy(d_blocks_by_cost, blocks_by_cost_host.data(), blocks_by_cost_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
}

void DeviceRenderer::setPrimarySplit(uint32_t above) {
	order_policy.primary_split_above = above;
	order_policy.primary_split_waves = 0xFFFFFFFFu;  // (asked for by name: however many tiles that is)
	if (!scene_ready || blocks_by_cost_host.empty())
		return;
	useDevice();
	synchronize();
	orderBlocksByCost();
	uploadBlocksByCost();
	frame_ready = false;
	++scene_version;
}



void DeviceRenderer::installOrder(const std::vector<uint32_t> &order, const std::array<std::array<uint32_t, 3>, XCD_GROUPS> &constants,
                                  const std::array<uint32_t, XCD_GROUPS> &splits) {
	order_host = order;
	queue_static = constants;
	split_tiles = splits;
	OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
	OCRT_HIP(hipMemcpy(d_order, order_host.data(), order_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	orderPrimaryBlocks();
	OCRT_HIP(hipMemcpy(d_primary_order, primary_order_host.data(), primary_order_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	OCRT_HIP(hipMemcpy(d_order_need, order_need_host.data(), order_need_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	uploadBlocksByCost();
	FrameCounters fresh{};
	for (uint32_t g = 0; g < XCD_GROUPS; ++g) {
		fresh.queue[g].work_tiles = constants[g][0];
		fresh.queue[g].cost_sum = constants[g][1];
		fresh.queue[g].hits = constants[g][2];
		fresh.queue[g].primary_blocks = primary_blocks[g];
		fresh.queue[g].split_units = std::min(split_tiles[g], constants[g][0]) * kp.ao_dirs;
		fresh.queue[g].head = fresh.queue[g].split_units;
	}
	OCRT_HIP(hipMemcpy(d_counters, &fresh, sizeof fresh, hipMemcpyHostToDevice));
	// (the frame count starts again at 0: no tile's flag may claim a frame)
	OCRT_HIP(hipMemsetAsync(d_tile_ready, 0, (tile_count ? tile_count : 1) * sizeof(uint32_t), (hipStream_t) stream));
	OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
	++scene_version;  // (nothing a captured frame bakes in has changed, but a frame in flight must not see the list change: callers synchronise)
}


#ifdef OCRT_PRIMARY_TICKS
extern void *primary_ticks_probe;  // (kernels.hip)
#endif
bool DeviceRenderer::measureTileCosts(unsigned frames) {
	if (!scene_ready)
		throw std::logic_error("measurement before upload");
	const bool has_ao = kp.ao_mode != AO_NONE && kp.ao_dirs > 0 && tile_count > 0 && tile_words.size() == tile_count;
	if (!has_ao || frames == 0)
		return false;
	useDevice();
	synchronize();
	void *d_cost = device_alloc(tile_count * sizeof(uint32_t));
	std::vector<uint32_t> ticks(tile_count);
	std::vector<float> sum(tile_count, 0.0f);
	try {
		for (unsigned f = 0; f <= frames; ++f) {  // (the first frame is not counted: code object pages, caches)
			OCRT_HIP(hipMemsetAsync(d_cost, 0, tile_count * sizeof(uint32_t), (hipStream_t) stream));
#ifdef OCRT_PRIMARY_TICKS
			primary_ticks_probe = d_cost;
#endif
			launchFrame(nullptr, nullptr, nullptr, d_cost);
#ifdef OCRT_PRIMARY_TICKS
			primary_ticks_probe = nullptr;
#endif
			OCRT_HIP(hipMemcpyAsync(ticks.data(), d_cost, tile_count * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t) stream));
			OCRT_HIP(hipStreamSynchronize((hipStream_t) stream));
			if (f > 0)
				for (size_t t = 0; t < tile_count; ++t)
					sum[t] += (float) ticks[t];
		}
	} catch (...) {
		device_free(d_cost);
		throw;
	}
	device_free(d_cost);
	tile_cost = std::move(sum);
	orderTiles();
	frame_ready = false;
	return true;
}

void DeviceRenderer::takeOrderFrom(const DeviceRenderer &other) {
	if (&other == this || other.tile_count != tile_count || other.order_host.empty())
		return;
	useDevice();
	synchronize();
	tile_words = other.tile_words;
	tile_cost = other.tile_cost;
	installOrder(other.order_host, other.queue_static, other.split_tiles);
}

void DeviceRenderer::tileOrder(std::vector<uint32_t> &order, std::vector<uint32_t> &constants, std::vector<uint32_t> &words, std::vector<float> &cost) const {
	order = order_host;
	constants.clear();
	for (const auto &q : queue_static)
		constants.insert(constants.end(), q.begin(), q.end());
	words = tile_words;
	cost = tile_cost;
}

void DeviceRenderer::setTileOrder(const std::vector<uint32_t> &order, const std::vector<uint32_t> &constants) {
	if (!scene_ready || order.size() != order_host.size() || constants.size() != 3 * XCD_GROUPS)
		throw std::invalid_argument("setTileOrder: a list of another frame");
	// every entry must be a tile of this frame (the kernel indexes the hit list's bases with it) and every group's count
	// must stay inside its segment
	for (uint32_t entry : order)
		if ((entry & 0x03FFFFFFu) >= tile_count && tile_count)
			throw std::invalid_argument("setTileOrder: no such tile");
	std::array<std::array<uint32_t, 3>, XCD_GROUPS> c{};
	const uint32_t strips = (kp.tiles_x + kp.strip_tiles - 1u) / kp.strip_tiles;
	for (uint32_t g = 0; g < XCD_GROUPS; ++g) {
		c[g] = { constants[3 * g], constants[3 * g + 1], constants[3 * g + 2] };
		if (c[g][0] > ((strips + XCD_GROUPS - 1u - g) >> 3) * kp.strip_tiles * kp.local_tile_rows)
			throw std::invalid_argument("setTileOrder: more tiles than the group's segment holds");
	}
	useDevice();
	synchronize();
	installOrder(order, c);
}

void DeviceRenderer::setOrderPolicy(float heavy, float runway, float split_above) {
	order_policy.heavy = heavy;
	order_policy.runway = runway;
	if (split_above >= 0.0f)
		order_policy.split_above = split_above;
	if (scene_ready && orderIsMeasured()) {
		useDevice();
		synchronize();
		orderTiles();
	}
}

}  // namespace ocrt

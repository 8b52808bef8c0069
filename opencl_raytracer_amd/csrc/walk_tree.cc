#include "walk_tree.h"

#include <algorithm>
#include <cmath>
#include <future>
#include <limits>

namespace ocrt {

namespace {

struct Box {
	float lo[3], hi[3];
	void reset() {
		for (int k = 0; k < 3; ++k) {
			lo[k] = std::numeric_limits<float>::infinity();
			hi[k] = -std::numeric_limits<float>::infinity();
		}
	}
	void grow(const float *l, const float *h) {
		for (int k = 0; k < 3; ++k) {
			lo[k] = std::fmin(lo[k], l[k]);
			hi[k] = std::fmax(hi[k], h[k]);
		}
	}
	double area() const {
		const double dx = (double) hi[0] - lo[0], dy = (double) hi[1] - lo[1], dz = (double) hi[2] - lo[2];
		return dx < 0.0 ? 0.0 : 2.0 * (dx * dy + dy * dz + dz * dx);
	}
};

struct Leaf {
	float lo[3], hi[3];
	float centre[3];
	uint32_t index;  // the reference's leaf number
};

constexpr int BINS = 16;

struct Builder {
	std::vector<Leaf> leaves;
	std::vector<NodeRec> out;  // 2n - 1 nodes: a subtree over m leaves occupies exactly 2m - 1 consecutive slots

	// Writes the subtree over leaves[begin, end) in pre-order from slot `at` on and returns its box.
	// The two halves of the top levels are built by separate threads (their slots are known in advance).
	Box build(size_t begin, size_t end, size_t at, int depth) {
		Box box;
		if (end - begin == 1) {
			const Leaf &l = leaves[begin];
			box.reset();
			box.grow(l.lo, l.hi);
			for (int k = 0; k < 3; ++k) {
				out[at].lo[k] = l.lo[k];
				out[at].hi[k] = l.hi[k];
			}
			out[at].skip = 1;
			out[at].leaf = l.index;
			return box;
		}
		// (a run of lopsided splits must not exhaust the stack: halve by index from depth 48 on; two leaves need no
		// binning to be told apart -- half of all inner nodes are such pairs, and the 48 bin boxes cost more than they do)
		const size_t mid = end - begin == 2 ? begin + 1 : depth < 48 ? split(begin, end) : begin + (end - begin) / 2;
		const size_t left_at = at + 1, right_at = at + 1 + (2 * (mid - begin) - 1);
		Box left, right;
		if (depth < 6 && end - begin > 2048) {  // (up to 64 tasks: the top of the tree is where the long passes are)
			std::future<Box> other = std::async(std::launch::async, [&] { return build(begin, mid, left_at, depth + 1); });
			right = build(mid, end, right_at, depth + 1);
			left = other.get();
		} else {
			left = build(begin, mid, left_at, depth + 1);
			right = build(mid, end, right_at, depth + 1);
		}
		box = left;
		box.grow(right.lo, right.hi);
		for (int k = 0; k < 3; ++k) {
			out[at].lo[k] = box.lo[k];
			out[at].hi[k] = box.hi[k];
		}
		out[at].skip = (uint32_t) (2 * (end - begin) - 1);
		out[at].leaf = 0xFFFFFFFFu;
		return box;
	}

	// Partitions leaves[begin, end) at the cheapest of the 3 x 15 bin boundaries; equal halves by
	// index when the centres do not separate.
	size_t split(size_t begin, size_t end) {
		// one pass: the centres' bounds, the box of all leaves and the largest leaf
		float c_lo[3], c_hi[3];
		for (int k = 0; k < 3; ++k) {
			c_lo[k] = std::numeric_limits<float>::infinity();
			c_hi[k] = -std::numeric_limits<float>::infinity();
		}
		Box all;
		all.reset();
		double largest = -1.0;
		size_t giant = begin;
		for (size_t i = begin; i < end; ++i) {
			const Leaf &l = leaves[i];
			for (int k = 0; k < 3; ++k) {
				c_lo[k] = std::fmin(c_lo[k], l.centre[k]);
				c_hi[k] = std::fmax(c_hi[k], l.centre[k]);
			}
			all.grow(l.lo, l.hi);
			const double dx = (double) l.hi[0] - l.lo[0], dy = (double) l.hi[1] - l.lo[1], dz = (double) l.hi[2] - l.lo[2];
			const double a = dx < 0.0 ? 0.0 : 2.0 * (dx * dy + dy * dz + dz * dx);
			if (a > largest) {
				largest = a;
				giant = i;
			}
		}
		// A leaf more than half as large as the whole node (a ground plane under a small model) would
		// drag every box it stays in up to its own size; centre-based bins cannot set it apart and the
		// greedy cost does not see far enough to want to.  It gets a node of its own.
		if (end - begin > 2 && largest >= 0.5 * all.area()) {
			std::swap(leaves[begin], leaves[giant]);
			return begin + 1;
		}
		// one more pass: the leaves into the bins of all three axes
		float scale[3];
		bool usable[3];
		Box bin_box[3][BINS];
		size_t bin_count[3][BINS] = { { 0 } };
		for (int axis = 0; axis < 3; ++axis) {
			const float extent = c_hi[axis] - c_lo[axis];
			usable[axis] = extent > 0.0f && std::isfinite(extent);
			scale[axis] = usable[axis] ? (float) BINS / extent : 0.0f;
			for (Box &b : bin_box[axis])
				b.reset();
		}
		for (size_t i = begin; i < end; ++i) {
			const Leaf &l = leaves[i];
			for (int axis = 0; axis < 3; ++axis) {
				if (!usable[axis])
					continue;
				const int b = bin_of(l.centre[axis], c_lo[axis], scale[axis]);
				bin_box[axis][b].grow(l.lo, l.hi);
				++bin_count[axis][b];
			}
		}
		double best_cost = std::numeric_limits<double>::infinity();
		int best_axis = -1, best_bin = 0;
		float best_scale = 0.0f;
		for (int axis = 0; axis < 3; ++axis) {
			if (!usable[axis])
				continue;
			double right_area[BINS];
			size_t right_count[BINS];
			Box sweep;
			sweep.reset();
			size_t n = 0;
			for (int b = BINS - 1; b >= 1; --b) {
				if (bin_count[axis][b])
					sweep.grow(bin_box[axis][b].lo, bin_box[axis][b].hi);
				n += bin_count[axis][b];
				right_area[b] = sweep.area();
				right_count[b] = n;
			}
			sweep.reset();
			n = 0;
			for (int b = 1; b < BINS; ++b) {  // split between bins b-1 and b
				if (bin_count[axis][b - 1])
					sweep.grow(bin_box[axis][b - 1].lo, bin_box[axis][b - 1].hi);
				n += bin_count[axis][b - 1];
				if (n == 0 || right_count[b] == 0)
					continue;
				const double cost = sweep.area() * (double) n + right_area[b] * (double) right_count[b];
				if (cost < best_cost) {
					best_cost = cost;
					best_axis = axis;
					best_bin = b;
					best_scale = scale[axis];
				}
			}
		}
		if (best_axis < 0)
			return begin + (end - begin) / 2;
		const float origin = c_lo[best_axis];
		const auto middle = std::partition(leaves.begin() + (long) begin, leaves.begin() + (long) end, [&](const Leaf &l) {
			return bin_of(l.centre[best_axis], origin, best_scale) < best_bin;
		});
		const size_t mid = (size_t) (middle - leaves.begin());
		return (mid == begin || mid == end) ? begin + (end - begin) / 2 : mid;
	}

	static int bin_of(float centre, float origin, float scale) {
		const int b = (int) ((centre - origin) * scale);
		return b < 0 ? 0 : b >= BINS ? BINS - 1 : b;
	}
};

}  // namespace

namespace {
double node_area(const NodeRec &n) {
	const double dx = (double) n.hi[0] - n.lo[0], dy = (double) n.hi[1] - n.lo[1], dz = (double) n.hi[2] - n.lo[2];
	return 2.0 * (dx * dy + dy * dz + dz * dx);
}
}  // namespace

double tree_cost(const std::vector<NodeRec> &nodes) {
	if (nodes.empty())
		return 0.0;
	const double root = node_area(nodes[0]);
	if (!(root > 0.0) || !std::isfinite(root))
		return std::numeric_limits<double>::infinity();
	struct Open {
		size_t end;
		double area;
	};
	std::vector<Open> parents;
	double sum = root;  // the root itself
	for (size_t i = 0; i < nodes.size(); ++i) {
		while (!parents.empty() && parents.back().end <= i)
			parents.pop_back();
		if (!parents.empty())
			sum += parents.back().area;
		if (nodes[i].skip > 1)
			parents.push_back({ i + nodes[i].skip, node_area(nodes[i]) });
	}
	return sum / root;
}

std::vector<NodeRec> contract_walk_tree(const std::vector<NodeRec> &nodes, double threshold) {
	struct Open {
		size_t end;  // first input index past the subtree
		size_t at;   // where the node went in the output
		double area;
	};
	std::vector<NodeRec> out;
	out.reserve(nodes.size());
	std::vector<Open> parents;
	auto close = [&](size_t upto) {
		while (!parents.empty() && parents.back().end <= upto) {
			out[parents.back().at].skip = (uint32_t) (out.size() - parents.back().at);
			parents.pop_back();
		}
	};
	for (size_t i = 0; i < nodes.size(); ++i) {
		close(i);
		const NodeRec &n = nodes[i];
		if (n.skip > 1 && !parents.empty() && node_area(n) > threshold * parents.back().area)
			continue;  // its children now answer to its parent
		out.push_back(n);
		if (n.skip > 1)
			parents.push_back({ i + n.skip, out.size() - 1, node_area(n) });
	}
	close(nodes.size());
	return out;
}

std::vector<NodeRec> nearest_children_first(const std::vector<NodeRec> &nodes, const double eye[3]) {
	std::vector<NodeRec> out;
	out.reserve(nodes.size());
	struct Child {
		double outside, centre;  // squared distances from the eye: to the box, to its centre
		size_t at;
	};
	// (an explicit stack of subtrees still to be written, children pushed farthest first: a degenerate array may be as
	// deep as it is long)
	std::vector<size_t> todo;
	std::vector<Child> children;
	if (!nodes.empty())
		todo.push_back(0);
	while (!todo.empty()) {
		const size_t i = todo.back();
		todo.pop_back();
		out.push_back(nodes[i]);  // (its skip count is that of its subtree: the same nodes, whatever their order)
		if (nodes[i].skip <= 1)
			continue;
		children.clear();
		for (size_t c = i + 1; c < i + nodes[i].skip && c < nodes.size(); c += nodes[c].skip ? nodes[c].skip : 1) {
			Child ch{ 0.0, 0.0, c };
			for (int k = 0; k < 3; ++k) {
				const double lo = nodes[c].lo[k], hi = nodes[c].hi[k];
				const double d = eye[k] < lo ? lo - eye[k] : eye[k] > hi ? eye[k] - hi : 0.0;
				const double m = 0.5 * lo + 0.5 * hi - eye[k];
				ch.outside += d * d;
				ch.centre += m * m;
			}
			children.push_back(ch);
		}
		std::stable_sort(children.begin(), children.end(), [](const Child &a, const Child &b) {
			return a.outside < b.outside || (a.outside == b.outside && a.centre < b.centre);
		});
		for (size_t k = children.size(); k-- > 0;)
			todo.push_back(children[k].at);
	}
	return out;
}

std::vector<NodeRec> rebuild_walk_tree(const std::vector<NodeRec> &packed) {
	Builder b;
	b.leaves.reserve((packed.size() + 1) / 2);
	for (const NodeRec &n : packed) {
		if (n.skip != 1)
			continue;
		Leaf l;
		for (int k = 0; k < 3; ++k) {
			l.lo[k] = n.lo[k];
			l.hi[k] = n.hi[k];
			l.centre[k] = 0.5f * n.lo[k] + 0.5f * n.hi[k];
		}
		l.index = n.leaf;
		b.leaves.push_back(l);
	}
	if (b.leaves.empty())
		return {};
	b.out.resize(2 * b.leaves.size() - 1);
	b.build(0, b.leaves.size(), 0, 0);
	return std::move(b.out);
}

}  // namespace ocrt

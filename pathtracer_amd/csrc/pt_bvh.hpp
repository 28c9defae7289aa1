// pt_bvh.hpp -- bounding-volume hierarchy for the extend stage (the reference is
// brute force, kernels.cu:62-68).  The BVH is only a CULLING structure: every triangle that reaches
// a leaf goes through the same ray_triangle() as the brute-force loop, and hits are ranked by
// (t, original triangle index), so the result is the brute-force result bit for bit -- provided no
// box test ever culls a triangle the fp32 Moeller-Trumbore test would accept.
//
// Why that needs care: ray_triangle accepts when the COMPUTED u, v satisfy 0<=u, 0<=v, u+v<=1, and
// those carry rounding error amplified by f = 1/det.  The reference's own culling gives the handle:
// an accepted hit has det >= EPSILON (Functions.cuh:72).  Term by term, with eps = 2^-24 the unit roundoff, D = |d||e0||e1|
// and det_true = D sin(phi) |cos(theta)| (phi between the edges, theta between ray and normal):
//     cross product   |dh| <= 2.83 eps |d||e1|           three-term dot product   <= 3 eps |a||b|          s = o - v0   <= eps |s|
//     det             |ddet| <= 5.83 eps D               numerators of u, v, t    <= 6.83 eps |s| D / |e0| (resp. |e1|, |d|)
//     |du| |e0|  <=  eps (6.83 |s| + 5.83 |e0|) D / det_true,    |dv| |e1| likewise with |e1|,    |dt| |d| likewise with |t||d|
// The true ray/plane intersection of an accepted hit is the computed (u, v) point -- inside the triangle -- moved by
// du e0 + dv e1, so it lies within |du||e0| + |dv||e1| <= 13.7 eps S D / det_true of the triangle, S >= |origin - vertex| +
// edge length (and |dt||d| obeys the same bound: |s| + |t||d| <= 2 S).  An accepted hit has det_true >= EPSILON - |ddet| >=
// 0.65 EPSILON for D <= 1, hence
//     delta_i = kappa_i * S,   kappa_i = 24 * eps * |e0||e1| / EPSILON        (13.7 / 0.65 = 21; |d| <= 1 for every ray this tracer makes)
// (round 1 and most of round 2 ran with 16: the bound for ONE of u, v with head-room -- no test ever found a hit outside
// it, rounding errors do not line up, but the claim is a bound; triangles with |e0||e1| >> 1 get boxes larger than the
// scene, which is what their unbounded error needs).  Each child box stores the max kappa of its subtree; the traversal inflates the box
// by kappa * S + a relative slack for its own arithmetic, and widens the ray interval by the root's
// delta.  Small triangles (the 10^4..10^5-triangle scenes this is for) get small inflations
// (|e0||e1| = 2e-4 -> delta = 3e-4 * S); large ones get large boxes, which only costs time.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "pt_hip.h"

namespace ptbvh {

#ifndef PT_BVH_LEAF
#define PT_BVH_LEAF 2                              // triangles per leaf, at most 8 (the leaf code keeps count - 1 in 3 bits); measured: pt_device.hpp
#endif
constexpr uint32_t kLeafSize = PT_BVH_LEAF;        // == ptd::kBvhLeafSize (pt_device.hpp: the traversal stack is sized from it)
static_assert(kLeafSize >= 1 && kLeafSize <= 8, "leaf code holds count - 1 in 3 bits");
constexpr float kEps32 = 5.9604645e-8f;            // 2^-24
constexpr float kKappaScale = 24.0f * kEps32 / 0.000001f;

// What the builder works with: the two children's boxes live in the parent.
struct Node {
	float lo0[3], hi0[3];
	float lo1[3], hi1[3];
	int32_t child0, child1;     // >= 0: node index; < 0: leaf, ~child = (first << 3) | (count - 1)
	float kappa0, kappa1;       // max kappa of the child's subtree
};
static_assert(sizeof(Node) == 64, "BVH node is 4 float4");

// 32 bytes, what the traversal reads: the same node with its boxes on a 16-bit grid over the scene's bounding cube
// (ptd::bvh_grid), lower corners rounded down, upper corners up, so a quantised box contains the exact one.  The node
// walk is bound by the number of cache lines a wave's 64 lanes touch per step (every lane is at a different node):
// two 16-byte loads per lane and step instead of four.  A word holds one axis of one child, {lo, hi}, so that the traversal
// picks the plane a ray enters through and the one it leaves through with one byte-permute each (pt_device.hpp box_hit).  The per-child kappa is gone from the node: every box is
// inflated for the scene's largest kappa, which is what the deepest boxes -- most of the tests -- needed anyway.
struct QNode {
	uint16_t box0[3][2];        // child 0: per axis {lo, hi}: one 32-bit word per axis, lo in its low half
	int32_t child0;             // as Node: a child's box and its reference share one 16-byte half of the node
	uint16_t box1[3][2];        // child 1
	int32_t child1;
};
static_assert(sizeof(QNode) == 32, "quantised BVH node is 2 x 16 bytes");

// ---- PT_BVH_WIDE (round 3; default on): four children per node ----
// The binary tree collapsed pairwise at pack time: a wide node holds the GRANDCHILDREN of a binary node (a child that is a leaf
// stays as it is), 2..4 children, each a 16-byte half {box, reference} as in QNode -- one 64-byte line per node step, and half as
// many dependent steps from the root to a leaf (13.2 -> 6.8 node steps per ray at 10 092 triangles).  Its inner depth is
// ceil(binary depth / 2); a walk defers up to three children per level (ptd::bvh_stack_total: the binary budget's worth of levels in
// LDS, the rest in private memory).  Empty slots hold kEmptyChild behind an inside-out box.
#ifndef PT_BVH_WIDE
#define PT_BVH_WIDE 1
#endif
constexpr int32_t kEmptyChild = 0x7FFFFFFD;        // == ptd::kBvhEmpty
struct Node4 {
	float lo[4][3], hi[4][3];
	int32_t child[4];
	uint32_t n;
};
struct QNode4 {
	struct Half { uint16_t box[3][2]; int32_t child; } c[4];
};
static_assert(sizeof(QNode4) == 64, "wide quantised BVH node is 4 x 16 bytes");

// 64 bytes: what the leaf test needs, in BVH order.  {v0.xyz, e0.x} {e0.yz, e1.xy} {e1.z, orig index bits, 0, 0} {pad}
struct LeafTri { float f[16]; };

// ---- normal classes: a tight inflation for rays that graze nothing ----
// The worst case kappa_i * S above (det barely above EPSILON) belongs to a GRAZING ray: det_true = |d| |e0||e1| sin(phi)
// |cos(theta)| (phi between the edges, theta between ray and normal).  If |cos(theta)| >= rho and sin(phi) >= kSinFloor
// the same term-by-term bound reads
//     delta <= 13.7 * eps * S / (rho * sin(phi))  <=  kKappaRho / rho * S,     kKappaRho = 16 * eps / kSinFloor
// whatever the triangle's size (no EPSILON in it: det_true is bounded directly; the computed determinant's own relative
// error e = 5.83 eps / (rho sin(phi)) scales the bound by 1 / (1 - e): 1.4 % at rho = kRhoMin, inside the 16-for-13.7 head-room,
// which would last down to rho = 1e-5) -- at rho = 0.02 that is 1.9e-4 * S against
// 1e-2 * S for the quads of the 10 092-triangle scene, whose boxes the worst case makes half as wide again.
// So the builder sorts the triangles' planes (normals, sign-agnostic) into at most kMaxClasses classes, each a cone of
// half-angle h around an axis, stored as {axis * cos(h), sin(h)} (with margins): for a ray at angle beta to the plane
// perpendicular to the axis, sin(beta) = |dhat . axis|, and its angle to any plane of the class is at least beta - h, so
//     |cos(theta)| >= sin(beta - h) >= |dhat . axis| cos(h) - sin(h)  =: rho_k      (one dot product and a subtraction)
// for EVERY triangle of the class.  In a scene whose triangles all belong to a class, rho = min_k rho_k bounds how close
// to grazing a ray comes to anything, and the ray inflates every box by
//     min(worst case, kKappaRho / rho * S),     kKappaRho = 16 * eps / kSinFloor
// -- graded, not a two-way choice: in an axis-aligned room 94 % of the rays have rho >= 0.02 (1.9e-4 * S or less), 4.5 % lie
// between 0.005 and 0.02 (at most 8e-4 * S), and only what lies below kRhoMin (3 rays in 10 000) keeps the worst case -- which on scenes
// of a few hundred LARGE triangles is larger than the boxes themselves (300 triangles: 0.15 * S), so every ray that fell
// back to it tested the whole scene, holding its wave for the time of twenty (measured with the cut at 1e-3, 3 rays in
// 1000: the fused engine ran 192..588 triangles at 4.0..6.5 Grays/s between 9.1 at 108 and 9.6 at 972).  Rays of a scene with slivers or more plane directions than classes keep the worst case.
constexpr float kRhoMin = 1e-4f;                   // == ptd::kBvhRhoMin: below this |cos(ray, normal)| bound a ray counts as grazing (worst case)
constexpr float kSinFloor = 0.25f;                 // triangles whose edges e0, e1 are closer to parallel than this are never classified
constexpr float kKappaRho = 16.0f * kEps32 / kSinFloor;      // == ptd::kBvhKappaRho
constexpr uint32_t kMaxClasses = 8;                // == ptd::kBvhMaxClasses
constexpr double kClassAngle = 0.14;               // radians (8 degrees): how far a normal may be from its class's axis

struct NormalClass { float axis[3], sin_h; };      // 16 bytes, what the traversal reads: axis * cos(h) (rounded down), sin(h) (rounded up)
// The displacement du e0 + dv e1 lies IN the triangle's plane: along a coordinate axis a it is at most |delta| sin(angle(normal, a)),
// so a box need not grow along its triangles' normal at all.  Per class and axis: in_plane[a] = sin(min(90 deg, angle(class
// axis, a) + h)) (rounded up) -- the walls of an axis-aligned room inflate within their own plane only, and a ray that
// grazes them (the one with the large delta) is pruned by the slab across it.
struct ClassBlock {                                // 320 bytes at the end of the BVH part of the blob
	uint32_t n_classes, all_classified;
	float kappa_free;                              // largest worst-case kappa among the triangles of no class (0: none)
	uint32_t second_root;                          // node index of the SECOND tree's root (0: one tree only), see build()
	NormalClass cls[kMaxClasses];
	float in_plane[kMaxClasses][4];                // {x, y, z, the class's largest worst-case kappa}
	uint32_t first_is_free;                        // 1: the tree at node 0 holds free triangles (no triangle has a class)
	uint32_t n_first;                              // leaf records [0, n_first) belong to the tree at node 0, the rest to the second
	float free_max_edge;                           // longest edge among the free triangles
	float free_center[3], free_radius;             // a sphere around the free triangles' vertices: |origin - vertex| <= |origin - centre| + radius
	uint32_t _pad2[5];
};
static_assert(sizeof(ClassBlock) == 320, "class block is 320 bytes");

struct Built {
	ClassBlock classes{};
	std::vector<Node> nodes;
	std::vector<Node4> nodes4;                     // PT_BVH_WIDE: the same trees collapsed (classes.second_root then indexes THIS array)
	std::vector<LeafTri> tris;
	float center[3] = { 0, 0, 0 };
	float radius = 0.0f;
	float kappa_root = 0.0f;
	float max_edge = 0.0f;
};

struct Prim { float lo[3], hi[3], c[3]; float kappa; uint32_t idx; };

// Greedy classification of the triangles' unit normals n (double, sign-agnostic), LARGEST kappa first; sliver[i]: never
// classified.  A triangle that finds no class (a sliver, or a ninth plane direction) is "free": it keeps its own worst-case
// kappa, and the largest of those, kappa_free, is what every box is inflated by for them -- so a room of large quads with a
// finely tessellated object in it gets classes for the quads (whose worst case would be larger than the room) and a small
// kappa_free for the object, instead of the worst case of the largest triangle for every ray as long as one triangle
// stayed unclassified.
inline ClassBlock classify_normals(const std::vector<double>& n, const std::vector<char>& sliver, const std::vector<float>& kappa,
                                   std::vector<int>* class_of = nullptr) {
	ClassBlock out{};
	const size_t count = sliver.size();
	if (class_of) class_of->assign(count, -1);         // -1: free
	double axes[kMaxClasses][3];
	double cos_h[kMaxClasses];
	float kappa_of[kMaxClasses];                   // the largest worst-case kappa among a class's members: what a ray that grazes the class falls back to
	uint32_t k_used = 0;
	bool all = true;
	float kappa_free = 0.0f;
	const double cos_accept = std::cos(kClassAngle);
	std::vector<uint32_t> order(count);
	std::iota(order.begin(), order.end(), 0u);
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return kappa[a] > kappa[b]; });
	for (size_t oi = 0; oi < count; oi++) {
		const size_t i = order[oi];
		if (sliver[i] || !(kappa[i] == kappa[i])) { all = false; kappa_free = std::max(kappa_free, kappa[i] == kappa[i] ? kappa[i] : INFINITY); continue; }
		const double* ni = &n[3 * i];
		bool placed = false;
		for (uint32_t k = 0; k < k_used && !placed; k++) {
			double c = std::fabs(ni[0] * axes[k][0] + ni[1] * axes[k][1] + ni[2] * axes[k][2]);
			// (a much smaller triangle joins only where it does not widen the cone: a wall's class must not come to graze every ray
			// within kClassAngle of the wall for the sake of a facet that is cheap to leave free)
			if (c >= cos_accept && (kappa[i] >= 0.25f * kappa_of[k] || c >= cos_h[k])) {
				cos_h[k] = std::min(cos_h[k], c); kappa_of[k] = std::max(kappa_of[k], kappa[i]); placed = true;
				if (class_of) (*class_of)[i] = (int)k;
			}
		}
		if (!placed) {
			// a new class only for triangles worth one: those of the size of the largest (the rest stay free, which costs their own small kappa)
			if (k_used == kMaxClasses || (k_used != 0 && kappa[i] < 0.25f * kappa_of[0])) { all = false; kappa_free = std::max(kappa_free, kappa[i]); continue; }
			for (int c = 0; c < 3; c++) axes[k_used][c] = ni[c];
			cos_h[k_used] = 1.0;
			kappa_of[k_used] = kappa[i];
			if (class_of) (*class_of)[i] = (int)k_used;
			k_used++;
		}
	}
	out.kappa_free = kappa_free;
	// (all_classified: no free triangle -- kept for readers of the blob; the traversal looks at kappa_free)
	out.n_classes = k_used;
	out.all_classified = (all && count > 0) ? 1u : 0u;
	for (uint32_t k = 0; k < k_used; k++) {
		// members were accepted against the double axis; the stored float axis differs by < 1e-7 rad, the 1e-3 rad below covers it
		double h = std::acos(std::min(1.0, cos_h[k])) + 1e-3;
		double alen = std::sqrt(axes[k][0] * axes[k][0] + axes[k][1] * axes[k][1] + axes[k][2] * axes[k][2]);
		// the traversal's dhat is good to ~1e-6 (v_rcp) and its dot product to a few ulp: 1e-4 relative on the axis and 1e-5
		// absolute on sin(h) keep the computed rho_k below the true |dhat . axis| cos(h) - sin(h)
		for (int c = 0; c < 3; c++) out.cls[k].axis[c] = (float)(axes[k][c] / alen * std::cos(h) * (1.0 - 1e-4));
		out.cls[k].sin_h = (float)(std::sin(h) * 1.0001 + 1e-5);
		for (int c = 0; c < 3; c++) {
			double to_axis = std::acos(std::min(1.0, std::fabs(axes[k][c]) / alen)) + h;          // a member's normal is at most this far from coordinate axis c ...
			out.in_plane[k][c] = (float)std::min(1.0, (to_axis >= 1.5707963267948966 ? 1.0 : std::sin(to_axis)) * 1.0001 + 1e-6);   // ... or exactly across it
		}
		out.in_plane[k][3] = kappa_of[k];
	}
	return out;
}

// x on the grid {glo + q * cell}: the largest q with glo + q*cell <= x (down) / the smallest with glo + q*cell >= x (up),
// in double against the float grid the device uses, clamped to 16 bits (a clamped corner only happens for the far-away
// box of a single-leaf root, which must not be entered anyway, or when x is outside the cube, which bvh_grid excludes)
inline uint16_t grid_down(float x, float glo, float cell) {
	double q = std::floor(((double)x - (double)glo) / (double)cell);
	q = q >= 0.0 ? std::min(q, 65535.0) : 0.0;                 // (NaN -> 0)
	while (q > 0.0 && (double)glo + q * (double)cell > (double)x) q -= 1.0;
	return (uint16_t)q;
}
inline uint16_t grid_up(float x, float glo, float cell) {
	double q = std::ceil(((double)x - (double)glo) / (double)cell);
	q = q <= 65535.0 ? std::max(q, 0.0) : 65535.0;             // (NaN -> 65535)
	while (q < 65535.0 && (double)glo + q * (double)cell < (double)x) q += 1.0;
	return (uint16_t)q;
}

inline std::vector<QNode> quantize(const std::vector<Node>& nodes, const float glo[3], float cell) {
	std::vector<QNode> out(nodes.size());
	for (size_t i = 0; i < nodes.size(); i++) {
		const Node& n = nodes[i];
		QNode& q = out[i];
		for (int k = 0; k < 3; k++) {
			q.box0[k][0] = grid_down(n.lo0[k], glo[k], cell); q.box0[k][1] = grid_up(n.hi0[k], glo[k], cell);
			q.box1[k][0] = grid_down(n.lo1[k], glo[k], cell); q.box1[k][1] = grid_up(n.hi1[k], glo[k], cell);
		}
		q.child0 = n.child0; q.child1 = n.child1;
	}
	return out;
}

// binary node `root2` and everything below it, appended to `out`; returns the wide node's index
inline int32_t collapse(const std::vector<Node>& nodes, int32_t root2, std::vector<Node4>& out) {
	const int32_t me = (int32_t)out.size();
	out.emplace_back();
	Node4 w{};
	auto add = [&](const float* lo, const float* hi, int32_t ref) {
		for (int k = 0; k < 3; k++) { w.lo[w.n][k] = lo[k]; w.hi[w.n][k] = hi[k]; }
		w.child[w.n++] = ref;
	};
	const Node& b = nodes[(size_t)root2];
	const int32_t refs[2] = { b.child0, b.child1 };
	for (int side = 0; side < 2; side++) {
		if (refs[side] >= 0) {
			const Node& c = nodes[(size_t)refs[side]];
			add(c.lo0, c.hi0, c.child0);
			add(c.lo1, c.hi1, c.child1);
		} else add(side ? b.lo1 : b.lo0, side ? b.hi1 : b.hi0, refs[side]);
	}
	for (uint32_t k = 0; k < w.n; k++) if (w.child[k] >= 0) w.child[k] = collapse(nodes, w.child[k], out);
	for (uint32_t k = w.n; k < 4; k++) { for (int a = 0; a < 3; a++) { w.lo[k][a] = 1e30f; w.hi[k][a] = -1e30f; } w.child[k] = kEmptyChild; }
	out[(size_t)me] = w;
	return me;
}

// the most entries a walk of the wide tree can have deferred at once: a node with k children defers up to k - 1 of them
inline uint32_t stack_need4(const std::vector<Node4>& nodes, uint32_t root = 0) {
	if (nodes.empty()) return 0;
	uint32_t worst = 0;
	std::vector<std::pair<int32_t, uint32_t>> todo{ { (int32_t)root, 0u } };
	while (!todo.empty()) {
		auto [i, held] = todo.back();
		todo.pop_back();
		const Node4& w = nodes[(size_t)i];
		const uint32_t here = held + (w.n - 1u);
		worst = std::max(worst, here);
		for (uint32_t k = 0; k < w.n; k++) if (w.child[k] >= 0) todo.push_back({ w.child[k], here });
	}
	return worst;
}

inline std::vector<QNode4> quantize4(const std::vector<Node4>& nodes, const float glo[3], float cell);

inline int32_t encode_leaf(uint32_t first, uint32_t count) { return ~(int32_t)((first << 3) | (count - 1)); }

inline std::vector<QNode4> quantize4(const std::vector<Node4>& nodes, const float glo[3], float cell) {
	std::vector<QNode4> out(nodes.size());
	for (size_t i = 0; i < nodes.size(); i++)
		for (int c = 0; c < 4; c++) {
			for (int k = 0; k < 3; k++) {
				out[i].c[c].box[k][0] = grid_down(nodes[i].lo[c][k], glo[k], cell);
				out[i].c[c].box[k][1] = grid_up(nodes[i].hi[c][k], glo[k], cell);
			}
			out[i].c[c].child = nodes[i].child[c];
		}
	return out;
}

inline void bounds_of(const std::vector<Prim>& p, uint32_t a, uint32_t b, float lo[3], float hi[3], float& kappa) {
	for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
	kappa = 0.0f;
	for (uint32_t i = a; i < b; i++) {
		for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[i].lo[k]); hi[k] = std::max(hi[k], p[i].hi[k]); }
		kappa = std::max(kappa, p[i].kappa);
	}
}

// Surface-area-heuristic split under a depth budget.  The traversal's stack (LDS, one deferred sibling per level) is
// sized from the triangle count alone -- ptd::bvh_stack_depth(n) = the balanced depth + kDepthSlack -- so the builder may
// unbalance the tree only as far as that allows: a subtree with `depth_left` levels to go holds at most
// kLeafSize << depth_left triangles, and a split is admissible if both sides still fit one level down.  Within that
// range the split minimises area(left) * n_left + area(right) * n_right over the three axes: a full sweep over the
// sorted centroids for ranges up to kSweepMax triangles, 32 bins per axis above.  The slack of three levels leaves
// the top of the tree unconstrained (the budget only bites where a subtree is already nearly full).
// PT_BVH_SAH 0: the median along the axis whose halves have the smallest total area (the tree the first measurements
// of DESIGN 4.4 were made with).
#ifndef PT_BVH_SAH
#define PT_BVH_SAH 1
#endif
constexpr uint32_t kSweepMax = 4096;
constexpr int kBins = 32;

inline float half_area(const float lo[3], const float hi[3]) {
	float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
	return dx * dy + dy * dz + dz * dx;
}

struct Box {
	float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
	void add(const Prim& q) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], q.lo[k]); hi[k] = std::max(hi[k], q.hi[k]); } }
	void add(const Box& q) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], q.lo[k]); hi[k] = std::max(hi[k], q.hi[k]); } }
	float area() const { return half_area(lo, hi); }
};

inline bool centroid_before(const Prim& x, const Prim& y, int axis) {
	return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.idx < y.idx);
}

// Reorders p[a, b) and returns the number of triangles that go left, within [min_left, max_left].
inline uint32_t choose_split(std::vector<Prim>& p, uint32_t a, uint32_t b, uint32_t min_left, uint32_t max_left) {
	const uint32_t n = b - a;
	float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (uint32_t i = a; i < b; i++)
		for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], p[i].c[k]); chi[k] = std::max(chi[k], p[i].c[k]); }
	float best = INFINITY;
	int best_axis = -1;
	uint32_t best_left = 0;
	int best_bin = -1;
	if (PT_BVH_SAH && n <= kSweepMax) {
		std::vector<float> right_area(n);
		for (int axis = 0; axis < 3; axis++) {
			std::sort(p.begin() + a, p.begin() + b, [axis](const Prim& x, const Prim& y) { return centroid_before(x, y, axis); });
			Box r;
			for (uint32_t i = n; i-- > 1;) { r.add(p[a + i]); right_area[i] = r.area(); }       // right_area[i]: triangles i..n-1
			Box l;
			for (uint32_t i = 1; i < n; i++) {
				l.add(p[a + i - 1]);
				if (i < min_left || i > max_left) continue;
				float cost = l.area() * (float)i + right_area[i] * (float)(n - i);
				if (cost < best) { best = cost; best_axis = axis; best_left = i; }
			}
		}
	} else if (PT_BVH_SAH) {
		for (int axis = 0; axis < 3; axis++) {
			float width = chi[axis] - clo[axis];
			if (!(width > 0.0f) || !std::isfinite(width)) continue;
			Box bins[kBins];
			uint32_t count[kBins] = {};
			float scale = (float)kBins / width;
			for (uint32_t i = a; i < b; i++) {
				int k = std::min(kBins - 1, std::max(0, (int)((p[i].c[axis] - clo[axis]) * scale)));
				bins[k].add(p[i]); count[k]++;
			}
			float right_area[kBins];
			Box r;
			for (int k = kBins - 1; k >= 1; k--) { r.add(bins[k]); right_area[k] = r.area(); }
			Box l;
			uint32_t nl = 0;
			for (int k = 0; k + 1 < kBins; k++) {
				l.add(bins[k]); nl += count[k];
				if (nl < min_left || nl > max_left || nl == 0 || nl == n) continue;
				float cost = l.area() * (float)nl + right_area[k + 1] * (float)(n - nl);
				if (cost < best) { best = cost; best_axis = axis; best_left = nl; best_bin = k; }
			}
		}
	}
	if (best_axis >= 0 && best_bin >= 0) {
		float lo = clo[best_axis], scale = (float)kBins / (chi[best_axis] - clo[best_axis]);
		int axis = best_axis, bin = best_bin;
		auto mid = std::partition(p.begin() + a, p.begin() + b, [=](const Prim& x) {
			return std::min(kBins - 1, std::max(0, (int)((x.c[axis] - lo) * scale))) <= bin; });
		return (uint32_t)(mid - (p.begin() + a));          // == best_left: the same expression binned them
	}
	if (best_axis >= 0) {
		int axis = best_axis;
		if (axis != 2) std::nth_element(p.begin() + a, p.begin() + a + best_left, p.begin() + b, [axis](const Prim& x, const Prim& y) { return centroid_before(x, y, axis); });
		return best_left;                                   // (axis 2 was sorted last: already in order)
	}
	// no admissible plane (all centroids equal, NaNs, or PT_BVH_SAH 0): the median, along the axis whose halves are smallest
	uint32_t left = std::min(std::max(n / 2, min_left), max_left);
	int axis = 0;
	best = INFINITY;
	for (int k = 0; k < 3; k++) {
		std::nth_element(p.begin() + a, p.begin() + a + left, p.begin() + b, [k](const Prim& x, const Prim& y) { return centroid_before(x, y, k); });
		Box l, r;
		for (uint32_t i = a; i < a + left; i++) l.add(p[i]);
		for (uint32_t i = a + left; i < b; i++) r.add(p[i]);
		float cost = l.area() + r.area();
		if (cost < best) { best = cost; axis = k; }
	}
	std::nth_element(p.begin() + a, p.begin() + a + left, p.begin() + b, [axis](const Prim& x, const Prim& y) { return centroid_before(x, y, axis); });
	return left;
}

inline int32_t build_range(std::vector<Prim>& p, uint32_t a, uint32_t b, uint32_t depth_left, Built& out) {
	if (b - a <= kLeafSize) return encode_leaf(a, b - a);
	// both sides must fit one level down: at most kLeafSize << (depth_left - 1) triangles each (depth_left >= 1 here, by
	// the same invariant one level up; at the root it is ptd::bvh_stack_depth)
	const uint64_t cap = (uint64_t)kLeafSize << std::min(depth_left - 1u, 40u);
	const uint32_t n = b - a;
	uint32_t min_left = (uint64_t)n > cap ? (uint32_t)(n - cap) : 1u;
	uint32_t max_left = (uint32_t)std::min<uint64_t>(cap, n - 1u);
	uint32_t mid = a + choose_split(p, a, b, std::max(min_left, 1u), max_left);
	uint32_t me = (uint32_t)out.nodes.size();
	out.nodes.emplace_back();
	Node nd{};
	bounds_of(p, a, mid, nd.lo0, nd.hi0, nd.kappa0);
	bounds_of(p, mid, b, nd.lo1, nd.hi1, nd.kappa1);
	nd.child0 = build_range(p, a, mid, depth_left - 1u, out);
	nd.child1 = build_range(p, mid, b, depth_left - 1u, out);
	out.nodes[me] = nd;
	return (int32_t)me;
}

// inner levels on the longest root-to-leaf path = the most deferred siblings a traversal can hold
inline uint32_t inner_depth(const std::vector<Node>& nodes, uint32_t root = 0) {
	if (nodes.empty()) return 0;
	uint32_t deepest = 0;
	std::vector<std::pair<int32_t, uint32_t>> todo{ { (int32_t)root, 1u } };
	while (!todo.empty()) {
		auto [i, d] = todo.back();
		todo.pop_back();
		deepest = std::max(deepest, d);
		if (nodes[(size_t)i].child0 >= 0) todo.push_back({ nodes[(size_t)i].child0, d + 1u });
		if (nodes[(size_t)i].child1 >= 0) todo.push_back({ nodes[(size_t)i].child1, d + 1u });
	}
	return deepest;
}

// stack_depth: ptd::bvh_depth_budget(n_tris), the binary levels the traversal's stack is sized for
inline Built build(const pt_vertex* v, const pt_triangle* t, uint32_t n_tris, uint32_t stack_depth) {
	Built out;
	std::vector<Prim> prims(n_tris);
	std::vector<double> normals((size_t)n_tris * 3);
	std::vector<char> sliver(n_tris);
	float slo[3] = { INFINITY, INFINITY, INFINITY }, shi[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (uint32_t i = 0; i < n_tris; i++) {
		const pt_float4* P[3] = { &v[t[i].vertexIdx0].position, &v[t[i].vertexIdx1].position, &v[t[i].vertexIdx2].position };
		Prim& pr = prims[i];
		for (int k = 0; k < 3; k++) { pr.lo[k] = INFINITY; pr.hi[k] = -INFINITY; }
		for (int c = 0; c < 3; c++) {
			const float xyz[3] = { P[c]->x, P[c]->y, P[c]->z };
			for (int k = 0; k < 3; k++) { pr.lo[k] = std::min(pr.lo[k], xyz[k]); pr.hi[k] = std::max(pr.hi[k], xyz[k]); }
		}
		for (int k = 0; k < 3; k++) {
			pr.c[k] = 0.5f * (pr.lo[k] + pr.hi[k]);
			slo[k] = std::min(slo[k], pr.lo[k]); shi[k] = std::max(shi[k], pr.hi[k]);
		}
		float e0[3] = { P[1]->x - P[0]->x, P[1]->y - P[0]->y, P[1]->z - P[0]->z };
		float e1[3] = { P[2]->x - P[0]->x, P[2]->y - P[0]->y, P[2]->z - P[0]->z };
		float l0 = std::sqrt(e0[0] * e0[0] + e0[1] * e0[1] + e0[2] * e0[2]);
		float l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
		pr.kappa = kKappaScale * l0 * l1;
		float e2[3] = { e1[0] - e0[0], e1[1] - e0[1], e1[2] - e0[2] };
		float l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
		out.max_edge = std::max(out.max_edge, std::max(l0, std::max(l1, l2)));
		pr.idx = i;
		{
			double a0[3] = { (double)P[1]->x - P[0]->x, (double)P[1]->y - P[0]->y, (double)P[1]->z - P[0]->z };
			double a1[3] = { (double)P[2]->x - P[0]->x, (double)P[2]->y - P[0]->y, (double)P[2]->z - P[0]->z };
			double nx = a0[1] * a1[2] - a0[2] * a1[1], ny = a0[2] * a1[0] - a0[0] * a1[2], nz = a0[0] * a1[1] - a0[1] * a1[0];
			double nl = std::sqrt(nx * nx + ny * ny + nz * nz);
			double ll = std::sqrt(a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2]) * std::sqrt(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
			sliver[i] = !(nl > 0.0) || !(ll > 0.0) || !std::isfinite(nl) || !std::isfinite(ll) || nl / ll < 1.02 * (double)kSinFloor;     // sin(phi) = |e0 x e1| / (|e0||e1|)
			normals[3 * (size_t)i + 0] = nl > 0.0 ? nx / nl : 0.0; normals[3 * (size_t)i + 1] = nl > 0.0 ? ny / nl : 0.0; normals[3 * (size_t)i + 2] = nl > 0.0 ? nz / nl : 0.0;
		}
	}
	std::vector<int> class_of;
	{
		std::vector<float> kappas(n_tris);
		for (uint32_t i = 0; i < n_tris; i++) kappas[i] = prims[i].kappa;
		out.classes = classify_normals(normals, sliver, kappas, &class_of);
	}
	float r2 = 0.0f;
	for (int k = 0; k < 3; k++) {
		out.center[k] = n_tris ? 0.5f * (slo[k] + shi[k]) : 0.0f;
		float h = n_tris ? 0.5f * (shi[k] - slo[k]) : 0.0f;
		r2 += h * h;
	}
	out.radius = std::sqrt(r2);
	out.nodes.reserve(n_tris);
	if (n_tris == 0) return out;
	// ---- two trees.  What a ray owes a CLASSIFIED triangle (graded by how close it comes to grazing the class's planes: up to the
	// class's own worst case, within the plane) and what it owes a FREE one (kappa_free, always, in every direction) are different
	// inflations, and in one tree every box pays the larger: round 2's room of large quads with a 40 000-facet ball ran at 0.1 Grays/s
	// because the few rays in 1000 that come within rho ~ 1e-3 of a wall inflated every box OF THE BALL by more than a facet's size.
	// So the classified triangles get a tree of their own (node 0) and the free ones another (ClassBlock::second_root); a ray walks
	// the first with the class inflation alone, then the second with kappa_free alone (pt_device.hpp: a marker on the traversal
	// stack switches the ray's constants between the two).  The classified tree goes first: it is the one of few, large
	// triangles -- the walls behind the detailed object --, cheap to walk, and the hit distance it leaves both prunes the walk of
	// the many free triangles and tightens S in their kappa_free * S (measured, room of 300 + ball of 2208 triangles, before S was
	// tightened: walls first 3.6 / 5.4 Grays/s fused / wavefront, ball first 3.2 / 5.2).  S for the free tree is further bounded
	// per ray by a sphere around the free triangles (free_center, free_radius) and their own longest edge: a ball of small
	// facets in a large room owes kappa_free times ITS distance from the ray's origin, not the room's size.
	// Hits are ranked by (t, original index) whatever the order of the
	// tests, so the result is still the loop over all triangles.  The marker takes one stack level: with two trees each is built
	// one level inside the budget.  A scene whose triangles are all classified, or all free, has one tree as before.
	std::stable_partition(prims.begin(), prims.end(), [&](const Prim& q) { return class_of[q.idx] >= 0; });
	uint32_t n_cls = 0;
	{
		float flo[3] = { INFINITY, INFINITY, INFINITY }, fhi[3] = { -INFINITY, -INFINITY, -INFINITY };
		for (uint32_t i = 0; i < n_tris; i++) {
			if (class_of[i] >= 0) { n_cls++; continue; }
			const pt_float4* P[3] = { &v[t[i].vertexIdx0].position, &v[t[i].vertexIdx1].position, &v[t[i].vertexIdx2].position };
			auto len = [](const pt_float4& p, const pt_float4& q) { float x = q.x - p.x, y = q.y - p.y, z = q.z - p.z; return std::sqrt(x * x + y * y + z * z); };
			out.classes.free_max_edge = std::max(out.classes.free_max_edge, std::max(len(*P[0], *P[1]), std::max(len(*P[0], *P[2]), len(*P[1], *P[2]))));
			for (int c = 0; c < 3; c++) {
				const float xyz[3] = { P[c]->x, P[c]->y, P[c]->z };
				for (int k = 0; k < 3; k++) { flo[k] = std::min(flo[k], xyz[k]); fhi[k] = std::max(fhi[k], xyz[k]); }
			}
		}
		if (n_cls != n_tris) {
			double rr = 0.0;
			for (int k = 0; k < 3; k++) { out.classes.free_center[k] = 0.5f * (flo[k] + fhi[k]); double h = 0.5 * ((double)fhi[k] - (double)flo[k]); rr += h * h; }
			out.classes.free_radius = (float)(std::sqrt(rr) * 1.0001 + 1e-6);       // (the centre's own rounding)
			if (!(out.classes.free_radius == out.classes.free_radius)) out.classes.free_radius = INFINITY;
		}
	}
	const bool two = n_cls != 0 && n_cls != n_tris;
	const uint32_t budget = two ? stack_depth - 1u : stack_depth;
	auto build_tree = [&](uint32_t a, uint32_t b) -> uint32_t {
		const uint32_t root_at = (uint32_t)out.nodes.size();
		int32_t root = build_range(prims, a, b, budget, out);
		if (root < 0) {                                   // a single leaf: wrap it so that a tree's root is always a node
			Node n{};
			bounds_of(prims, a, b, n.lo0, n.hi0, n.kappa0);
			n.child0 = root;
			for (int k = 0; k < 3; k++) { n.lo1[k] = 1e30f; n.hi1[k] = -1e30f; }  // inside-out box (65535..0 on the grid): (almost) never entered; entering it re-tests one triangle, which is harmless
			n.child1 = encode_leaf(a, 1);
			n.kappa1 = 0.0f;
			out.nodes.push_back(n);
		}
		return root_at;
	};
	if (two) {
		build_tree(0, n_cls);
		out.classes.second_root = build_tree(n_cls, n_tris);
	} else build_tree(0, n_tris);
	if (PT_BVH_WIDE) {
		collapse(out.nodes, 0, out.nodes4);
		if (two) {
			const uint32_t second2 = out.classes.second_root;
			out.classes.second_root = (uint32_t)out.nodes4.size();
			collapse(out.nodes, (int32_t)second2, out.nodes4);
		}
	}
	out.classes.first_is_free = n_cls == 0 ? 1u : 0u;
	out.classes.n_first = two ? n_cls : n_tris;
	for (uint32_t i = 0; i < n_tris; i++) out.kappa_root = std::max(out.kappa_root, prims[i].kappa == prims[i].kappa ? prims[i].kappa : INFINITY);
	out.tris.resize(n_tris);
	for (uint32_t k = 0; k < n_tris; k++) {
		uint32_t i = prims[k].idx;
		const pt_float4 &v0 = v[t[i].vertexIdx0].position, &v1 = v[t[i].vertexIdx1].position, &v2 = v[t[i].vertexIdx2].position;
		float* f = out.tris[k].f;
		std::memset(f, 0, sizeof(LeafTri));
		f[0] = v0.x; f[1] = v0.y; f[2] = v0.z;
		f[3] = v1.x - v0.x; f[4] = v1.y - v0.y; f[5] = v1.z - v0.z;     // the reference's edge0 (Functions.cuh:65)
		f[6] = v2.x - v0.x; f[7] = v2.y - v0.y; f[8] = v2.z - v0.z;     // edge1 (:66)
		std::memcpy(&f[9], &i, 4);
	}
	return out;
}

} // namespace ptbvh

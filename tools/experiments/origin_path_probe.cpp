// origin_path_probe.cpp -- how much of a bounced ray's BVH walk is spent on the ANCESTORS of the leaf it starts on?
// (measurement for the "start the walk at the leaf a ray leaves" idea, which was then built and lost: DESIGN 4.4; host build of
// the device code, tests/host/hip_shim)
//   g++ -O2 -std=c++17 -ffp-contract=off -Itests/host/hip_shim -Ipathtracer_amd/csrc -Iinclude -o /tmp/origin_path_probe tools/experiments/origin_path_probe.cpp
//   /tmp/origin_path_probe scene.blob
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "pt_device.hpp"

using namespace ptd;

int main(int argc, char** argv) {
	if (argc < 2) return 2;
	FILE* f = std::fopen(argv[1], "rb");
	std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
	std::vector<unsigned char> blob((size_t)n);
	if (std::fread(blob.data(), 1, (size_t)n, f) != (size_t)n) return 2;
	const SceneHeader* hdr = reinterpret_cast<const SceneHeader*>(blob.data());
	uint32_t n_tris = hdr->n_tris, n_nodes = hdr->n_nodes;
	float4 camera{ 0, 0, 0, 1 };
	BvhView bv = make_bvh_view(blob.data(), n_tris, hdr->n_mats, camera);
	const float4* tris = reinterpret_cast<const float4*>(blob.data() + sizeof(SceneHeader));
	// parents, and the inner node that holds each triangle's leaf
	std::vector<int32_t> parent(n_nodes, -1), leaf_parent(n_tris, -1);
	std::vector<int> depth(n_nodes, 0);
	for (uint32_t i = 0; i < n_nodes; i++) {
		uint4 w0 = bv.nodes[(size_t)i * 2], w1 = bv.nodes[(size_t)i * 2 + 1];
		int32_t c[2] = { (int32_t)w0.w, (int32_t)w1.w };        // half nodes {x, y, z, child}
		for (int k = 0; k < 2; k++) {
			if (c[k] >= 0) { parent[c[k]] = (int32_t)i; }
			else {
				uint32_t code = (uint32_t)~c[k], first = code >> 3, count = (code & 7u) + 1u;
				for (uint32_t t = 0; t < count; t++) {
					float4 r2 = bv.leaves[(size_t)(first + t) * 4 + 2];
					leaf_parent[__float_as_uint(r2.y)] = (int32_t)i;
				}
			}
		}
	}
	std::vector<uint32_t> stack_mem(bvh_stack_depth(n_tris) * 256);
	std::mt19937 gen(7);
	std::uniform_real_distribution<float> U(-1.0f, 1.0f);
	const int W = 240, H = 135;
	unsigned long long rays = 0, steps = 0, anc_steps = 0, leafs = 0, tri_tests = 0, anc_depth = 0, pops_after = 0;
	std::vector<char> is_anc(n_nodes);
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
		Vec3 o{ 0, 0, 0 };
		Vec3 d{ (x + 0.5f) / W * 2 - 1, ((y + 0.5f) / H * 2 - 1) * H / W, 1.0f };
		float len = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z + 1.0f);
		d = Vec3{ d.x / len, d.y / len, d.z / len };
		float best = FLT_MAX; uint32_t tri = kNoHit;
		bvh_intersect(bv, o, d, best, tri, stack_mem.data());
		if (tri == kNoHit) continue;
		// bounce: cube sample flipped into the hemisphere of the un-normalised normal (Functions.cuh:91-109 in spirit)
		float4 r0 = tris[4 * tri], r1 = tris[4 * tri + 1], r2 = tris[4 * tri + 2];
		Vec3 e0{ r0.w, r1.x, r1.y }, e1{ r1.z, r1.w, r2.x };
		Vec3 nrm = cross3(e0, e1);
		Vec3 nd{ U(gen), U(gen), U(gen) };
		if (dot3(nd, nrm) < 0) nd = Vec3{ -nd.x, -nd.y, -nd.z };
		Vec3 no{ o.x + d.x * best, o.y + d.y * best, o.z + d.z * best };
		float nbest = best;               // the stale hit distance
		uint32_t ntri = kNoHit;
		std::fill(is_anc.begin(), is_anc.end(), 0);
		int ad = 0;
		for (int32_t a = leaf_parent[tri]; a >= 0; a = parent[a]) { is_anc[a] = 1; ad++; }
		anc_depth += ad;
		// the walk of bvh_intersect, one lane, with counters
		BvhRay r = bvh_ray_setup(bv, no, nd, nbest);
		int32_t cur = 0; int sp = 0; uint32_t* stack = stack_mem.data();
		while (cur != kBvhDone) {
			if (cur >= 0) {
				steps++; anc_steps += is_anc[cur];
				const uint4* nn = bv.nodes + (size_t)cur * 2;
				uint4 w0 = nn[0], w1 = nn[1];
				int32_t c0 = (int32_t)w0.w, c1 = (int32_t)w1.w;
				float thi = nbest + r.dt + 1e-6f * nbest, tn0, tn1;
				bool h0 = box_hit(w0.x, w0.y, w0.z, r, thi, tn0), h1 = box_hit(w1.x, w1.y, w1.z, r, thi, tn1);
				if (h0 && h1) { bool f0 = tn0 <= tn1; stack[sp * 256] = (uint32_t)(f0 ? c1 : c0); sp++; cur = f0 ? c0 : c1; }
				else if (h0) cur = c0; else if (h1) cur = c1; else cur = bvh_pop(stack, sp);
			} else {
				leafs++; tri_tests += ((uint32_t)~cur & 7u) + 1u;
				bvh_leaf(bv.leaves, (uint32_t)~cur, bv.det_bounded, no, nd, nbest, ntri);
				cur = bvh_pop(stack, sp);
			}
		}
		rays++;
	}
	std::printf("%u triangles, %u nodes: %llu bounced rays; per ray: node steps %.2f, of them on ancestors of the origin leaf %.2f (that leaf sits %.2f levels deep), "
	            "leaves %.2f, triangle tests %.2f\n", n_tris, n_nodes, rays, (double)steps / rays, (double)anc_steps / rays, (double)anc_depth / rays,
	            (double)leafs / rays, (double)tri_tests / rays);
	(void)pops_after;
	return 0;
}

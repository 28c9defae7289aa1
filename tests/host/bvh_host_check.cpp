// BVH traversal == brute force, checked on the CPU with the DEVICE code itself: pt_device.hpp is compiled by g++ through
// tests/host/hip_shim (a wave of one lane) and every ray is intersected twice -- intersect_range over all triangles
// (the reference's loop, kernels.cu:62-68) and bvh_intersect -- against a scene blob packed by the library.  The blob is
// copied into an exact-size heap block so that AddressSanitizer sees any read outside it.
//   bvh_host_check <blob file> <rays file: N x {ox oy oz dx dy dz best_t} float32> <camera x y z>
// prints "rays N mismatches M" and exits 1 on any mismatch.
#include <cstdio>
#include <cstdlib>
#include <vector>

#define PT_BVH_TRACK_DEPTH
#include "pt_device.hpp"

using namespace ptd;

static std::vector<unsigned char> read_file(const char* path) {
	std::vector<unsigned char> v;
	FILE* f = std::fopen(path, "rb");
	if (!f) { std::perror(path); std::exit(2); }
	std::fseek(f, 0, SEEK_END);
	long n = std::ftell(f);
	std::fseek(f, 0, SEEK_SET);
	v.resize((size_t)n);
	if (n && std::fread(v.data(), 1, (size_t)n, f) != (size_t)n) { std::perror("read"); std::exit(2); }
	std::fclose(f);
	return v;
}

int main(int argc, char** argv) {
	if (argc < 6) { std::fprintf(stderr, "usage: bvh_host_check blob rays cx cy cz\n"); return 2; }
	std::vector<unsigned char> blob_v = read_file(argv[1]), rays_v = read_file(argv[2]);
	unsigned char* blob = (unsigned char*)std::malloc(blob_v.size());          // exact size: ASan guards both ends
	std::memcpy(blob, blob_v.data(), blob_v.size());
	const SceneHeader* hdr = reinterpret_cast<const SceneHeader*>(blob);
	if (hdr->magic != kSceneMagic || hdr->blob_bytes != blob_v.size() || hdr->n_nodes == 0) { std::fprintf(stderr, "not a BVH scene blob\n"); return 2; }
	float4 camera{ (float)std::atof(argv[3]), (float)std::atof(argv[4]), (float)std::atof(argv[5]), 1.0f };
	uint32_t n_tris = hdr->n_tris, n_mats = hdr->n_mats;
	const float4* tris = reinterpret_cast<const float4*>(blob + sizeof(SceneHeader));
	BvhView bv = make_bvh_view(blob, n_tris, n_mats, camera);
	size_t depth = bvh_stack_depth(n_tris);
	uint32_t* stack = (uint32_t*)std::malloc(depth * 256 * sizeof(uint32_t));      // [level][256 lanes], lane 0 used
	std::memset(stack, 0xDB, depth * 256 * sizeof(uint32_t));                       // LDS is not zero on the device
	const float* rays = reinterpret_cast<const float*>(rays_v.data());
	size_t n_rays = rays_v.size() / (7 * sizeof(float)), bad = 0, hits = 0;
	for (size_t k = 0; k < n_rays; k++) {
		const float* r = rays + 7 * k;
		Vec3 o{ r[0], r[1], r[2] }, d{ r[3], r[4], r[5] };
		float best_a = r[6], best_b = r[6];
		uint32_t tri_a = kNoHit, tri_b = kNoHit;
		intersect_range(tris, 0, n_tris, bv.det_bounded, o, d, best_a, tri_a);
		bvh_intersect(bv, o, d, best_b, tri_b, stack);
		hits += tri_a != kNoHit;
		if (tri_a != tri_b || __float_as_uint(best_a) != __float_as_uint(best_b)) {
			if (bad < 10) std::fprintf(stderr, "ray %zu: brute force (%u, %.9g) BVH (%u, %.9g)\n", k, tri_a, best_a, tri_b, best_b);
			bad++;
		}
	}
	std::printf("rays %zu hits %zu mismatches %zu deepest %d lds_levels %zu\n", n_rays, hits, bad, g_bvh_deepest, depth);
	std::free(stack);
	std::free(blob);
	return bad ? 1 : 0;
}

// The extend stage of a BVH scene as a WAVE on the CPU: extend_bvh_block (pt_device.hpp, what k_extend<false, true>
// runs) executed by 64 host threads, one per lane, through tests/host/hip_shim -- ballots, shuffles and the refill
// counter behave as on the device, so the persistent loop, the lane refill and the quorum rule of the node walk are
// exercised with rays that diverge.  Every queue slot's {t, triangle} is compared with the loop over all triangles.
//   extend_wave_check <blob file> <rays file: N x {ox oy oz dx dy dz best_t} float32> <camera x y z> [lanes=64] [first=0]
// first = 1: the bounce-0 form (k_extend<true, true>): every ray starts at the camera with best_t = FLT_MAX, the queue
// holds directions only; the rays file's origins and distances are ignored.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

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
	if (argc < 6) { std::fprintf(stderr, "usage: extend_wave_check blob rays cx cy cz [lanes]\n"); return 2; }
	std::vector<unsigned char> blob_v = read_file(argv[1]), rays_v = read_file(argv[2]);
	unsigned char* blob = (unsigned char*)std::malloc(blob_v.size());          // exact size: ASan guards both ends
	std::memcpy(blob, blob_v.data(), blob_v.size());
	const SceneHeader* hdr = reinterpret_cast<const SceneHeader*>(blob);
	if (hdr->magic != kSceneMagic || hdr->blob_bytes != blob_v.size() || hdr->n_nodes == 0) { std::fprintf(stderr, "not a BVH scene blob\n"); return 2; }
	float4 camera{ (float)std::atof(argv[3]), (float)std::atof(argv[4]), (float)std::atof(argv[5]), 1.0f };
	const int lanes = argc > 6 ? std::atoi(argv[6]) : 64;
	const bool first = argc > 7 && std::atoi(argv[7]) != 0;
	const uint32_t n_tris = hdr->n_tris, n_mats = hdr->n_mats;
	const float4* tris = reinterpret_cast<const float4*>(blob + sizeof(SceneHeader));
	const BvhView bv = make_bvh_view(blob, n_tris, n_mats, camera);
	const float* rays = reinterpret_cast<const float*>(rays_v.data());
	const uint32_t count = (uint32_t)(rays_v.size() / (7 * sizeof(float)));
	// one block (G = 1, blockIdx 0): entry j sits in slot j
	std::vector<float4> qa(count), qb(count);
	std::vector<uint2> hit(count, uint2{ 0xDEADBEEFu, 0xDEADBEEFu });
	for (uint32_t k = 0; k < count; k++) {
		const float* r = rays + 7 * k;
		qa[k] = float4{ r[0], r[1], r[2], r[3] };
		qb[k] = first ? float4{ r[3], r[4], r[5], 0.0f } : float4{ r[4], r[5], r[6], 0.0f };
	}
	std::vector<uint32_t> s_stack((size_t)bvh_stack_depth(n_tris) * 256, 0xDEADBEEFu);      // LDS is not zero on the device: a slot read before it is written must show
	uint32_t s_next = 0;
	WaveShim wave;
	wave.lanes = lanes;
	std::vector<std::thread> th;
	for (int l = 0; l < lanes; l++)
		th.emplace_back([&, l] {
			threadIdx.x = (unsigned)l;
			t_wave = &wave;
			if (first) extend_bvh_block<true>(bv, camera, nullptr, qb.data(), count, count, hit.data(), s_stack.data(), &s_next);
			else extend_bvh_block<false>(bv, camera, qa.data(), qb.data(), count, count, hit.data(), s_stack.data(), &s_next);
		});
	for (auto& t : th) t.join();
	size_t bad = 0, hits = 0;
	for (uint32_t k = 0; k < count; k++) {
		const float* r = rays + 7 * k;
		float best = first ? FLT_MAX : r[6];
		uint32_t tri = kNoHit;
		Vec3 o = first ? Vec3{ camera.x, camera.y, camera.z } : Vec3{ r[0], r[1], r[2] };
		intersect_range(tris, 0, n_tris, bv.det_bounded, o, Vec3{ r[3], r[4], r[5] }, best, tri);
		hits += tri != kNoHit;
		if (hit[k].y != tri || hit[k].x != __float_as_uint(best)) {
			if (bad < 10) std::fprintf(stderr, "slot %u: brute force (%u, %.9g) wave (%u, %.9g)\n", k, tri, best, hit[k].y, __uint_as_float(hit[k].x));
			bad++;
		}
	}
	std::printf("rays %u hits %zu mismatches %zu lanes %d\n", count, hits, bad, lanes);
	std::free(blob);
	return bad ? 1 : 0;
}

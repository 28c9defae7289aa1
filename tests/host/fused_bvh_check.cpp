// The fused engine's wave-level forms on the CPU.  Scenes without a BVH: fused_samples_out_of_step (k_fused<*, *, true>), a wave of
// 64 host threads per 64 pixels.  BVH scenes: fused_bvh_block (pt_fused_bvh.hpp, what k_fused_bvh runs) executed block by block,
// each block as 256 host threads in four waves of 64 through tests/host/hip_shim -- the persistent walk, the service of
// finished rays at a quorum, pixels handed out from the block's LDS counter, walks that start at a leaf (path lists) all
// run as on the device.  Arrays have exactly the size the library gives them, LDS stand-ins start from garbage.
//   fused_bvh_check <blob> <pt_frame file> <frame_first> <mean_first> <n_samples> <pixels per block> <out file> [lanes per wave]
// out = pixel_count x float4 mean, pixel_count x uint32 rng; prints the ray count.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "pt_fused_bvh.hpp"

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
static uint32_t steps(uint32_t x, unsigned long long k) { while (k--) x = xs_step(x); return x; }
template <typename T> static T* exact(size_t n) {
	T* p = (T*)std::malloc((n ? n : 1) * sizeof(T));
	std::memset(p, 0xDB, (n ? n : 1) * sizeof(T));
	return p;
}

int main(int argc, char** argv) {
	if (argc < 8) { std::fprintf(stderr, "usage: fused_bvh_check blob frame frame_first mean_first n_samples pixels_per_block out [lanes]\n"); return 2; }
	std::vector<unsigned char> blob_v = read_file(argv[1]), frame_v = read_file(argv[2]);
	if (frame_v.size() != sizeof(pt_frame)) { std::fprintf(stderr, "frame file is %zu bytes, pt_frame is %zu\n", frame_v.size(), sizeof(pt_frame)); return 2; }
	pt_frame fr;
	std::memcpy(&fr, frame_v.data(), sizeof(fr));
	const uint32_t frame_first = (uint32_t)std::atol(argv[3]), mean_first = (uint32_t)std::atol(argv[4]), n_samples = (uint32_t)std::atol(argv[5]);
	// pixels per block: a number = contiguous ranges of that many; iG = interleaved tiles over G blocks (fused_bvh_range)
	const bool interleaved = argv[6][0] == 'i';
	const uint32_t range = interleaved ? 0u : (uint32_t)std::atol(argv[6]);
	const int lanes = argc > 8 ? std::atoi(argv[8]) : 64;                  // < 64: partial waves (a block of 4 x lanes threads)
	unsigned char* blob = (unsigned char*)std::malloc(blob_v.size());
	std::memcpy(blob, blob_v.data(), blob_v.size());
	const SceneHeader* hdr = reinterpret_cast<const SceneHeader*>(blob);
	if (hdr->magic != kSceneMagic || hdr->blob_bytes != blob_v.size()) { std::fprintf(stderr, "not a scene blob\n"); return 2; }
	const bool brute_force = hdr->n_nodes == 0;          // no BVH: the out-of-step form of k_fused (fused_samples_out_of_step), wave by wave
	const FrameDev f = to_dev(fr);
	const bool strided = fr.rng_mode == PT_RNG_STRIDED;
	const uint32_t N = f.pixel_count, n_tris = hdr->n_tris, n_mats = hdr->n_mats;
	float4* mean = exact<float4>(N);
	uint32_t* rng = exact<uint32_t>(N);
	for (uint32_t li = 0; li < N; li++) { mean[li] = float4{ 1.0f, 1.0f, 1.0f, 1.0f }; rng[li] = steps(f.seed + f.pixel_first + li, (f.pixel_first + li) >> 5); }
	uint32_t* s_stack = exact<uint32_t>((size_t)bvh_stack_depth(n_tris) * 256);
	uint32_t* s_jump = exact<uint32_t>(1024);
	uint32_t* s_first = exact<uint32_t>(32);
	uint32_t* s_per = exact<uint32_t>(32);
	static const ByteJump jump31 = make_byte_jump(31);
	for (uint32_t k = 0; k < 1024u; k++) s_jump[k] = (&jump31.t[0][0])[k];
	for (uint32_t b = 0; b < 32; b++) { s_first[b] = steps(1u << b, (unsigned long long)frame_first * f.stride); s_per[b] = steps(1u << b, f.stride); }
	uint32_t* s_stride = exact<uint32_t>(1024);                     // step^stride as the kernels build it
	for (unsigned t = 0; t < 256; t++) { threadIdx.x = t; build_stride_table(s_per, s_stride); }
	const float4* tris = reinterpret_cast<const float4*>(blob + sizeof(SceneHeader));
	unsigned long long rays = 0, diffuse = 0;
	if (brute_force) {
		// k_fused<*, *, true>: a block per 256 pixels, a lane per pixel; the camera-ray records as stage_primary_records makes them
		const bool use_prim = n_tris <= 1024;                                        // kPrimMaxTris (pt_kernels.hip)
		float4* prim = exact<float4>((size_t)2 * n_tris);
		for (unsigned t = 0; t < 256; t++) { threadIdx.x = t; stage_primary_records(tris, n_tris, Vec3{ f.origin.x, f.origin.y, f.origin.z }, prim); }
		const bool det_bounded = (hdr->flags & kSceneDetBounded) != 0;
		const uint32_t chunks = (N + 255u) / 256u;
		for (uint32_t blk = 0; blk < chunks; blk++) {
			WaveShim waves[4];
			for (auto& w : waves) w.lanes = 64;
			std::vector<unsigned long long> cnt(256 * 2, 0);
			std::vector<std::thread> th;
			for (unsigned t = 0; t < 256; t++)
				th.emplace_back([&, t] {
					threadIdx.x = t;
					t_wave = &waves[t / 64];
					const uint32_t li = blk * 256u + t;
					const bool in_range = li < N;
					const uint32_t i = f.pixel_first + (in_range ? li : 0u);
					uint32_t start = in_range ? rng[li] : 0u;
					if (strided) start = mat_apply_lds(s_first, start);
					uint32_t r = start, n_rays = 0, n_diffuse = 0;
					float4 m = in_range ? mean[li] : float4{ 0, 0, 0, 0 };
					FusedLdsCtx c{ f, tris, tris + (size_t)n_tris * 4, use_prim ? prim : nullptr, s_jump, s_stride, n_tris, det_bounded, mean_first, n_samples };
					if (strided) fused_samples_out_of_step<true>(c, in_range, i % (uint32_t)f.width, i / (uint32_t)f.width, start, r, m, n_rays, n_diffuse);
					else fused_samples_out_of_step<false>(c, in_range, i % (uint32_t)f.width, i / (uint32_t)f.width, start, r, m, n_rays, n_diffuse);
					if (in_range) { mean[li] = m; if (!strided) rng[li] = r; }
					cnt[2 * t] = n_rays; cnt[2 * t + 1] = n_diffuse;
				});
			for (auto& t : th) t.join();
			for (unsigned t = 0; t < 256; t++) { rays += cnt[2 * t]; diffuse += cnt[2 * t + 1]; }
		}
		FILE* out = std::fopen(argv[7], "wb");
		if (!out) { std::perror(argv[7]); return 2; }
		std::fwrite(mean, sizeof(float4), N, out);
		std::fwrite(rng, sizeof(uint32_t), N, out);
		std::fclose(out);
		std::printf("pixels %u samples %u rays %llu diffuse %llu blocks %u\n", N, n_samples, rays, diffuse, chunks);
		return 0;
	}
	const BvhView bv = make_bvh_view(blob, n_tris, n_mats, f.origin);
	const uint32_t G = interleaved ? (uint32_t)std::atol(argv[6] + 1) : (N + range - 1) / range;
	gridDim.x = G;
	for (uint32_t blk = 0; blk < G; blk++) {
		blockIdx.x = blk;
		uint32_t s_next = 0;
		std::memset(s_stack, 0xDB, (size_t)bvh_stack_depth(n_tris) * 256 * sizeof(uint32_t));
		WaveShim waves[4];
		for (auto& w : waves) w.lanes = lanes;
		std::vector<unsigned long long> cnt(256 * 2, 0);
		std::vector<std::thread> th;
		for (unsigned w = 0; w < 4; w++)
			for (int l = 0; l < lanes; l++)
				th.emplace_back([&, w, l] {
					threadIdx.x = w * 64 + (unsigned)l;
					t_wave = &waves[w];
					FusedBvhCtx c{ f, bv, tris, tris + (size_t)n_tris * 4, s_jump, s_first, s_stride, mean_first, n_samples, rng, mean };
					const FusedBvhRange mine = fused_bvh_range(N, range, blk, G);
					uint32_t n_rays = 0, n_diffuse = 0;
					if (strided) fused_bvh_block<true>(c, mine.first, mine.count, mine.tile_stride, s_stack, &s_next, n_rays, n_diffuse);
					else fused_bvh_block<false>(c, mine.first, mine.count, mine.tile_stride, s_stack, &s_next, n_rays, n_diffuse);
					cnt[2 * threadIdx.x] = n_rays; cnt[2 * threadIdx.x + 1] = n_diffuse;
				});
		for (auto& t : th) t.join();
		for (unsigned t = 0; t < 256; t++) { rays += cnt[2 * t]; diffuse += cnt[2 * t + 1]; }
	}
	FILE* out = std::fopen(argv[7], "wb");
	if (!out) { std::perror(argv[7]); return 2; }
	std::fwrite(mean, sizeof(float4), N, out);
	std::fwrite(rng, sizeof(uint32_t), N, out);
	std::fclose(out);
	std::printf("pixels %u samples %u rays %llu diffuse %llu blocks %u\n", N, n_samples, rays, diffuse, G);
	return 0;
}

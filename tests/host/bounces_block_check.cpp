// The wavefront engine's one-launch pass on the CPU: bounces_block (pt_wavefront.hpp, what k_bounces runs) executed block
// by block, each block as 256 host threads in four waves of 64 through tests/host/hip_shim -- ballot compaction into
// block-private queue slots, the LDS counters, the block barriers between bounces, pixel-owning blocks, the strided
// byte-table walk and the in-kernel fold into the mean all run as on the device.  Every array has exactly the size the
// library gives it (queue_capacity, pixels x samples), so AddressSanitizer sees any slot that falls outside.
//   bounces_block_check <blob> <pt_frame file> <frame_first> <n_samples> <samples_per_pass> <max blocks> <out file>
// out = pixel_count x float4 mean, pixel_count x uint32 rng; prints the ray count (sum of the per-bounce counters).
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "pt_wavefront.hpp"

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
// exact size (ASan guards both ends), filled with garbage: neither the workspace nor LDS is zero on the device, so a slot
// that is read before it is written must show.  n == 0: a valid pointer nobody may index.
template <typename T> static T* exact(size_t n) {
	T* p = (T*)std::malloc((n ? n : 1) * sizeof(T));
	std::memset(p, 0xDB, (n ? n : 1) * sizeof(T));
	return p;
}

int main(int argc, char** argv) {
	if (argc < 8) { std::fprintf(stderr, "usage: bounces_block_check blob frame frame_first n_samples samples_per_pass max_blocks out\n"); return 2; }
	std::vector<unsigned char> blob_v = read_file(argv[1]), frame_v = read_file(argv[2]);
	if (frame_v.size() != sizeof(pt_frame)) { std::fprintf(stderr, "frame file is %zu bytes, pt_frame is %zu\n", frame_v.size(), sizeof(pt_frame)); return 2; }
	pt_frame fr;
	std::memcpy(&fr, frame_v.data(), sizeof(fr));
	const uint32_t frame_first = (uint32_t)std::atol(argv[3]), n_samples = (uint32_t)std::atol(argv[4]);
	uint32_t spp = (uint32_t)std::atol(argv[5]);
	const uint32_t max_blocks = (uint32_t)std::atol(argv[6]);
	unsigned char* blob = (unsigned char*)std::malloc(blob_v.size());
	std::memcpy(blob, blob_v.data(), blob_v.size());
	const SceneHeader* hdr = reinterpret_cast<const SceneHeader*>(blob);
	if (hdr->magic != kSceneMagic || hdr->blob_bytes != blob_v.size() || hdr->n_nodes != 0) { std::fprintf(stderr, "not a brute-force scene blob\n"); return 2; }

	const FrameDev f = to_dev(fr);

	const bool sequential = fr.rng_mode == PT_RNG_SEQUENTIAL;
	if (spp == 0) spp = 1;                                                      // pt_hip_render_wavefront
	if (spp > n_samples) spp = n_samples;
	const uint32_t per_launch = spp;                                            // SEQUENTIAL: samples per launch, one in flight per pixel
	if (sequential) spp = 1;
	const uint32_t N = f.pixel_count, n_tris = hdr->n_tris, n_mats = hdr->n_mats;
	const uint32_t chunks = (N + 255u) / 256u;
	const uint32_t G = std::max(1u, std::min(chunks, max_blocks));             // grid_for(pixel_count) with max_blocks in place of 8 x CUs
	const size_t slots = (size_t)G * ((chunks + G - 1) / G) * spp * 256;        // queue_capacity
	const size_t paths_cap = (size_t)N * spp;
	float4 *qa[2], *qb[2], *qc[2]; uint32_t* qpid[2]; uint2* qhit[2];
	for (int k = 0; k < 2; k++) { qa[k] = exact<float4>(slots); qb[k] = exact<float4>(slots); qc[k] = exact<float4>(slots); qpid[k] = exact<uint32_t>(slots); qhit[k] = exact<uint2>(slots); }
	float4* radiance = exact<float4>(paths_cap);
	unsigned char* lit = exact<unsigned char>(paths_cap);
	uint32_t* ray_counts = exact<uint32_t>((size_t)(f.depth + 1) * G);
	std::memset(ray_counts, 0, (size_t)(f.depth + 1) * G * sizeof(uint32_t));      // the library zeroes them before a call's first pass
	uint32_t* pass_mat = exact<uint32_t>(64);
	float4* mean = exact<float4>(N);
	uint32_t* rng = exact<uint32_t>(N);
	uint32_t* first_pending = exact<uint32_t>(N);
	unsigned long long* need_masks = exact<unsigned long long>((size_t)chunks * 4);
	for (uint32_t li = 0; li < N; li++) { mean[li] = float4{ 1.0f, 1.0f, 1.0f, 1.0f }; rng[li] = steps(f.seed + f.pixel_first + li, (f.pixel_first + li) >> 5); }

	// the block's LDS
	const uint32_t nt4 = n_tris * 4, nm2 = n_mats * 2;
	float4* s_scene = exact<float4>((size_t)nt4 + nm2 + 2 * n_tris);
	uint32_t* s_jump = exact<uint32_t>(1024);
	uint32_t* s_stride = exact<uint32_t>(1024);
	uint32_t s_out = 0, s_traced = 0, s_round[2] = { 0xDBDBDBDBu, 0xDBDBDBDBu };
	static const ByteJump jump31 = make_byte_jump(31);
	const float4* g_tris = reinterpret_cast<const float4*>(blob + sizeof(SceneHeader));
	const float4* g_mats = g_tris + (size_t)n_tris * 4;
	const bool det_bounded = (hdr->flags & kSceneDetBounded) != 0;

	{   // k_need_masks, ahead of the first pass: one block here
		float4* s_prim = exact<float4>(128);
		gridDim.x = 1;
		blockIdx.x = 0;
		WaveShim block, waves[4];
		block.lanes = 256;
		for (auto& w : waves) w.lanes = 64;
		std::vector<std::thread> th;
		for (unsigned t = 0; t < 256; t++)
			th.emplace_back([&, t] {
				threadIdx.x = t;
				t_wave = &waves[t / 64];
				t_block = &block;
				need_masks_block(f, g_tris, n_tris, s_prim, need_masks, chunks * 4);
			});
		for (auto& t : th) t.join();
	}
	unsigned long long rays = 0;
	gridDim.x = G;
	for (uint32_t done = 0; done < n_samples; done += per_launch) {
		const uint32_t ns = std::min(per_launch, n_samples - done);
		if (!sequential)                                                        // k_pass_matrices
			for (uint32_t b = 0; b < 32; b++) {
				pass_mat[b] = steps(1u << b, (unsigned long long)(frame_first + done) * f.stride);
				pass_mat[32 + b] = steps(1u << b, f.stride);
			}
		for (uint32_t blk = 0; blk < G; blk++) {
			blockIdx.x = blk;
			WaveShim block, waves[4];
			block.lanes = 256;
			for (auto& w : waves) w.lanes = 64;
			std::vector<std::thread> th;
			for (unsigned t = 0; t < 256; t++)
				th.emplace_back([&, t] {
					threadIdx.x = t;
					t_wave = &waves[t / 64];
					t_block = &block;
					// k_bounces' prologue
					for (uint32_t k = t; k < 1024u; k += 256u) s_jump[k] = (&jump31.t[0][0])[k];
					if (!sequential) build_stride_table(pass_mat + 32, s_stride);
					for (uint32_t k = t; k < nt4; k += 256u) s_scene[k] = g_tris[k];
					for (uint32_t k = t; k < nm2; k += 256u) s_scene[nt4 + k] = g_mats[k];
					stage_primary_records(g_tris, n_tris, Vec3{ f.origin.x, f.origin.y, f.origin.z }, s_scene + nt4 + nm2);
					if (t == 0) { s_out = 0; s_traced = 0; s_round[0] = 0; s_round[1] = 0; }
					__syncthreads();
					ShadeCtx c{ f, s_scene, s_scene + nt4, s_jump, n_tris, det_bounded, (uint32_t)slots, radiance, lit, sequential ? rng : nullptr, &s_out, &s_traced };
					if (sequential)
						bounces_block<true>(c, s_scene + nt4 + nm2, s_stride, qa[0], qb[0], qc[0], qpid[0], qhit[0], qa[1], qb[1], qc[1], qpid[1], qhit[1],
						                    ray_counts, pass_mat, rng, ns, frame_first + done, mean, first_pending, need_masks, s_round);
					else
						bounces_block<false>(c, s_scene + nt4 + nm2, s_stride, qa[0], qb[0], qc[0], qpid[0], qhit[0], qa[1], qb[1], qc[1], qpid[1], qhit[1],
						                     ray_counts, pass_mat, rng, ns, frame_first + done, mean, first_pending, need_masks, s_round);
				});
			for (auto& t : th) t.join();
		}
	}
	for (uint32_t b = 0; b < f.depth; b++)                                         // k_fold_counts, stats[0], after the last pass
		for (uint32_t blk = 0; blk < G; blk++) rays += ray_counts[(size_t)b * G + blk];
	FILE* out = std::fopen(argv[7], "wb");
	if (!out) { std::perror(argv[7]); return 2; }
	std::fwrite(mean, sizeof(float4), N, out);
	std::fwrite(rng, sizeof(uint32_t), N, out);
	std::fclose(out);
	std::printf("pixels %u samples %u rays %llu blocks %u slots %zu\n", N, n_samples, rays, G, slots);
	return 0;
}

// The whole per-pixel recurrence on the CPU with the DEVICE functions themselves: pt_device.hpp is compiled by g++
// through tests/host/hip_shim (a wave of one lane, ASan + UBSan) and driven by a loop that mirrors k_fused
// (pathtracer_amd/csrc/pt_kernels.hip) line by line -- primary_direction (both division forms), the camera-ray records,
// intersect_range / bvh_intersect, shade_diffuse / shade_extended, running_mean, and the strided stream addressing.
// tests/test_device_code_on_host.py compares the image and the RNG streams with the oracle, bit for bit.
//   fused_host_check <blob file> <pt_frame file, 112 B> <frame_first> <mean_first> <n_samples> <out file>
// out = pixel_count x float4 mean, then pixel_count x uint32 rng.
#include <cstdio>
#include <cstdlib>
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

static uint32_t steps(uint32_t x, unsigned long long k) { while (k--) x = xs_step(x); return x; }

int main(int argc, char** argv) {
	if (argc < 7) { std::fprintf(stderr, "usage: fused_host_check blob frame frame_first mean_first n_samples out\n"); return 2; }
	std::vector<unsigned char> blob_v = read_file(argv[1]), frame_v = read_file(argv[2]);
	if (frame_v.size() != sizeof(pt_frame)) { std::fprintf(stderr, "frame file is %zu bytes, pt_frame is %zu\n", frame_v.size(), sizeof(pt_frame)); return 2; }
	pt_frame fr;
	std::memcpy(&fr, frame_v.data(), sizeof(fr));
	const uint32_t frame_first = (uint32_t)std::atol(argv[3]), mean_first = (uint32_t)std::atol(argv[4]), n_samples = (uint32_t)std::atol(argv[5]);
	unsigned char* blob = (unsigned char*)std::malloc(blob_v.size());          // exact size: ASan guards both ends
	std::memcpy(blob, blob_v.data(), blob_v.size());
	const SceneHeader* hdr = reinterpret_cast<const SceneHeader*>(blob);
	if (hdr->magic != kSceneMagic || hdr->blob_bytes != blob_v.size()) { std::fprintf(stderr, "not a scene blob\n"); return 2; }
	blockDim.x = 1;                                                            // block-cooperative staging loops cover everything

	const FrameDev f = to_dev(fr);

	const uint32_t n_tris = hdr->n_tris, n_mats = hdr->n_mats;
	const bool BVH = hdr->n_nodes != 0, STRIDED = fr.rng_mode == PT_RNG_STRIDED, extended = fr.shading == PT_SHADING_EXTENDED;
	const bool det_bounded = (hdr->flags & kSceneDetBounded) != 0;
	const float4* tris = reinterpret_cast<const float4*>(blob + sizeof(SceneHeader));
	const float4* mats = tris + (size_t)n_tris * 4;
	static const ByteJump jump31 = make_byte_jump(31);
	const uint32_t* s_jump = &jump31.t[0][0];
	BvhView bv{};
	if (BVH) bv = make_bvh_view(blob, n_tris, n_mats, f.origin);
	std::vector<uint32_t> stack((size_t)bvh_stack_depth(n_tris) * 256);
	std::vector<float4> prim((size_t)2 * n_tris + 1);
	const bool use_prim = !BVH && n_tris <= 1024;                              // kPrimMaxTris
	if (use_prim) stage_primary_records(tris, n_tris, Vec3{ f.origin.x, f.origin.y, f.origin.z }, prim.data());

	std::vector<float4> mean(f.pixel_count, float4{ 1.0f, 1.0f, 1.0f, 1.0f });   // the plugin's initial image; overwritten at mean index 0
	std::vector<uint32_t> rng_out(f.pixel_count);
	unsigned long long n_rays = 0;
	for (uint32_t li = 0; li < f.pixel_count; li++) {
		uint32_t i = f.pixel_first + li;
		uint32_t px = i % (uint32_t)f.width, py = i / (uint32_t)f.width;
		uint32_t start = steps(f.seed + i, i >> 5);                            // k_rng_seed
		if (STRIDED) start = steps(start, (unsigned long long)frame_first * f.stride);
		uint32_t rng = start;
		float4 m = mean[li];
		unsigned long long need = ~0ull;
		if (use_prim) need = primary_need_mask(f, px, py, true, tris, prim.data(), n_tris);
		for (uint32_t s = 0; s < n_samples; s++) {
			Vec3 o{ f.origin.x, f.origin.y, f.origin.z };
			Vec3 d = primary_direction(f, px, py, rng);
			float best = FLT_MAX;
			float4 step = float4{ 1.0f, 1.0f, 1.0f, 1.0f };
			float4 frame = float4{ 0.0f, 0.0f, 0.0f, 1.0f };
			uint32_t draws = 2;
			for (uint32_t b = 0; b < f.depth; b++) {
				n_rays++;
				uint32_t tri = kNoHit;
				if (extended) best = FLT_MAX;
				if (BVH) bvh_intersect(bv, o, d, best, tri, stack.data());
				else if (b == 0 && use_prim) intersect_range_primary(tris, prim.data(), n_tris, det_bounded, d, best, tri, need);
				else intersect_range(tris, 0, n_tris, det_bounded, o, d, best, tri);
				if (tri == kNoHit) break;
				float4 r2 = tris[4 * tri + 2], r3 = tris[4 * tri + 3];
				uint32_t mi = __float_as_uint(r3.x);
				float4 head = mats[2 * mi];
				uint32_t type = __float_as_uint(head.x);
				float4 color = mats[2 * mi + 1];
				if (extended && type != PT_MAT_EMISSIVE) {
					if (shade_extended(type, Vec3{ r2.y, r2.z, r2.w }, color, head.y, best, o, d, step, rng)) draws += 2;
				} else if (type == PT_MAT_DIFFUSE) {
					shade_diffuse(Vec3{ r2.y, r2.z, r2.w }, color, best, s_jump, o, d, step, rng);
					draws += 33;
				} else {
					if (type == PT_MAT_EMISSIVE)
						frame = float4{ step.x * color.x, step.y * color.y, step.z * color.z, step.w * color.w };
					break;
				}
			}
			m = running_mean(frame, m, mean_first + s);
			if (STRIDED) {
				if (draws != f.stride) rng = steps(start, f.stride);
				start = rng;
			}
		}
		mean[li] = m;
		rng_out[li] = rng;
	}
	FILE* out = std::fopen(argv[6], "wb");
	if (!out) { std::perror(argv[6]); return 2; }
	std::fwrite(mean.data(), sizeof(float4), mean.size(), out);
	std::fwrite(rng_out.data(), sizeof(uint32_t), rng_out.size(), out);
	std::fclose(out);
	std::printf("pixels %u samples %u rays %llu\n", f.pixel_count, n_samples, n_rays);
	std::free(blob);
	return 0;
}

// multi_gpu_host.cpp -- a C++ host that spreads ONE frame over several plugin instances, the way a maintainer of the
// reference's host (PathTracer/Program.cpp:51-72, one IPathTracer*) would go multi-GPU: one instance per GPU, each
// driven through the IPathTracer vtable from a thread of its own, then one combine.  Sample split (PT_SPLIT_SAMPLE,
// include/pt_plugin.h): the image does not depend on the number of instances.
//   multi_gpu_host <instances> <width> <height> <frames> <samples_per_frame> <out.raw> [tile]
// Instance k runs on device k % (visible devices): on a one-GPU box all of them share device 0.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "CameraData.h"
#include "IPathTracer.h"
#include "pt_plugin.h"

extern "C" IPathTracer* Create(unsigned int, int, int, const CameraData*);
extern "C" void Destroy(IPathTracer*);

static int die(const char* what) { std::fprintf(stderr, "%s: %s\n", what, pt_last_error()); return 1; }

int main(int argc, char** argv) {
	if (argc < 7) { std::fprintf(stderr, "usage: multi_gpu_host instances width height frames samples_per_frame out.raw [tile]\n"); return 2; }
	const uint32_t G = (uint32_t)std::atoi(argv[1]);
	const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), frames = std::atoi(argv[4]);
	const uint32_t spf = (uint32_t)std::atoi(argv[5]);
	const bool tile = argc > 7 && std::strcmp(argv[7], "tile") == 0;
	int devices = 0;
	if (pt_hip_device_count(&devices) != PT_OK || devices < 1) return die("no device");
	pt_camera_data cam = { { 0, 0, 0, 1 }, { 0, 0, 0, 1 }, 90.0f };
	std::vector<pt_handle*> inst(G, nullptr);
	for (uint32_t g = 0; g < G; g++) {
		inst[g] = pt_create_on_device((int)(g % (uint32_t)devices), w, h, &cam);
		if (!inst[g]) return die("pt_create_on_device");
		pt_options o;
		if (pt_get_options(inst[g], &o)) return die("pt_get_options");
		o.split_mode = tile ? PT_SPLIT_TILE : PT_SPLIT_SAMPLE;
		o.split_rank = g; o.split_ranks = G;
		o.samples_per_begin = spf;
		if (!tile) { o.rng_mode = PT_RNG_STRIDED; o.stride = PT_DEFAULT_STRIDE; }
		if (pt_set_options(inst[g], &o)) return die("pt_set_options");
	}
	// one host thread per instance, each calling its instance through the reference's vtable
	std::vector<std::thread> th;
	for (uint32_t g = 0; g < G; g++)
		th.emplace_back([&, g] {
			IPathTracer* p = static_cast<IPathTracer*>(inst[g]);
			CameraData c;
			for (int f = 0; f < frames; f++) { p->BeginDrawing(); p->Update(&c); p->FinalizeDrawing(); }
		});
	for (auto& t : th) t.join();
	std::vector<float> image((size_t)w * h * 4);
	if (tile) {
		size_t at = 0;
		for (uint32_t g = 0; g < G; g++) {
			uint32_t first = 0, count = 0;
			pt_get_size(inst[g], nullptr, nullptr, &first, &count);
			if ((size_t)first * 4 != at) { std::fprintf(stderr, "tiles do not abut\n"); return 1; }
			if (pt_readback(inst[g], image.data() + at, (size_t)count * 4)) return die("pt_readback");
			at += (size_t)count * 4;
		}
		if (at != image.size()) { std::fprintf(stderr, "tiles do not cover the image\n"); return 1; }
	} else {
		if (pt_combine_instances(inst.data(), G)) return die("pt_combine_instances");
		if (pt_readback(inst[0], image.data(), image.size())) return die("pt_readback");
	}
	unsigned long long rays = 0;
	for (uint32_t g = 0; g < G; g++) { pt_stats st; if (pt_get_stats(inst[g], &st)) return die("pt_get_stats"); rays += st.rays; }
	for (uint32_t g = 0; g < G; g++) Destroy(static_cast<IPathTracer*>(inst[g]));
	FILE* out = std::fopen(argv[6], "wb");
	if (!out) { std::perror(argv[6]); return 2; }
	std::fwrite(image.data(), sizeof(float), image.size(), out);
	std::fclose(out);
	std::printf("instances %u devices %d rays %llu\n", G, devices, rays);
	return 0;
}

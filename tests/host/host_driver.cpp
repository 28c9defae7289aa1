// host_driver.cpp -- a headless stand-in for the reference's host application loop
// (PathTracer/Program.cpp:51-91, 227-252): loads a plugin by name through
// IPathTracerManagement.h, creates it, runs `frames` iterations of
//     BeginDrawing(); Update(camera); FinalizeDrawing();
// exactly in the reference's order, optionally changes the camera / resizes mid-run, reads the
// running mean back through the plugin's pt_readback export and writes it as raw float32.
// Used by tests (C1 plumbing, camera-restart and resize semantics) for both the HIP plugin and
// the CPU-oracle plugin.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>

#include "IPathTracerManagement.h"

typedef int (*ReadbackFunc)(void*, float*, size_t);

int main(int argc, char** argv) {
	if (argc < 6) {
		std::fprintf(stderr, "usage: %s <library> <width> <height> <frames> <out.raw> [--move-at F] [--resize-at F W H]\n", argv[0]);
		return 2;
	}
	const char* library = argv[1];
	int width = std::atoi(argv[2]), height = std::atoi(argv[3]), frames = std::atoi(argv[4]);
	const char* out = argv[5];
	int moveAt = -1, resizeAt = -1, newW = 0, newH = 0;
	for (int a = 6; a < argc; a++) {
		if (!std::strcmp(argv[a], "--move-at") && a + 1 < argc) moveAt = std::atoi(argv[++a]);
		else if (!std::strcmp(argv[a], "--resize-at") && a + 3 < argc) { resizeAt = std::atoi(argv[++a]); newW = std::atoi(argv[++a]); newH = std::atoi(argv[++a]); }
	}

	CreatePathTracerFunc CreatePathTracer = nullptr;
	DestroyPathTracerFunc DestroyPathTracer = nullptr;
	if (!LoadPathTracerLibrary(library, &CreatePathTracer, &DestroyPathTracer)) return 3;

	CameraData* cameraData = new CameraData();
	IPathTracer* pathTracer = CreatePathTracer(0, width, height, cameraData);
	if (!pathTracer) { std::fprintf(stderr, "Create returned null\n"); return 4; }

	for (int frame = 0; frame < frames; frame++) {
		if (frame == moveAt) cameraData->fieldOfView -= 5.0f;       // Program.cpp:293-295
		if (frame == resizeAt) { width = newW; height = newH; pathTracer->Resize(width, height); }   // Program.cpp:208-209
		pathTracer->BeginDrawing();                                  // Program.cpp:57
		pathTracer->Update(cameraData);                              // Program.cpp:58
		pathTracer->FinalizeDrawing();                               // Program.cpp:70
	}

	// the additive readback export lives in the same library
	void* self = dlopen(library, RTLD_NOW | RTLD_NOLOAD);
	if (!self) self = dlopen((std::string("lib") + library + ".so").c_str(), RTLD_NOW | RTLD_NOLOAD);
	ReadbackFunc readback = self ? (ReadbackFunc)dlsym(self, "pt_readback") : nullptr;
	if (!readback) { std::fprintf(stderr, "pt_readback not exported by %s\n", library); return 5; }
	std::vector<float> image((size_t)width * height * 4);
	if (readback(pathTracer, image.data(), image.size()) != 0) { std::fprintf(stderr, "pt_readback failed\n"); return 6; }
	FILE* f = std::fopen(out, "wb");
	if (!f || std::fwrite(image.data(), sizeof(float), image.size(), f) != image.size()) { std::fprintf(stderr, "cannot write %s\n", out); return 7; }
	std::fclose(f);

	DestroyPathTracer(pathTracer);
	delete cameraData;
	if (self) dlclose(self);
	if (!UnloadPathTracerLibrary(library)) return 8;
	std::printf("host_driver: %d frames at %dx%d written to %s\n", frames, width, height, out);
	return 0;
}

// pt_headless -- the reference's host application without the window.
//
// What PathTracer/Program.cpp does with GLFW/GL, minus the display: load a plugin by name
// (Program.cpp:227-235), run the frame loop BeginDrawing / Update / FinalizeDrawing (Program.cpp:51-72),
// report frame time and fps the way the window title did (Program.cpp:78-90) -- plus rays/s from the
// plugin's counters --, optionally move the camera or resize mid-run (the key handlers of
// Program.cpp:254-298 and the resize callback :204-211), and write the running mean as a PFM image.
//
//   pt_headless [--plugin NAME] [--size WxH] [--frames N] [--spp-per-frame K] [--scene FILE] [--depth D]
//               [--engine fused|wavefront] [--shading reference|extended] [--fov DEG] [--pos x,y,z] [--move-at F] [--resize-at F WxH]
//               [--report-every N] [--out image.pfm]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>

#include "IPathTracerManagement.h"
#include "pt_plugin.h"

namespace {

template <class F> F sym(void* lib, const char* name) { return reinterpret_cast<F>(dlsym(lib, name)); }

bool write_pfm(const char* path, const std::vector<float>& rgba, int w, int h) {
	FILE* f = std::fopen(path, "wb");
	if (!f) return false;
	std::fprintf(f, "PF\n%d %d\n-1.0\n", w, h);           // little-endian float RGB, bottom row first -- the plugin's row order
	std::vector<float> rgb((size_t)w * h * 3);
	for (size_t i = 0; i < (size_t)w * h; i++) { rgb[3 * i] = rgba[4 * i]; rgb[3 * i + 1] = rgba[4 * i + 1]; rgb[3 * i + 2] = rgba[4 * i + 2]; }
	bool ok = std::fwrite(rgb.data(), sizeof(float), rgb.size(), f) == rgb.size();
	std::fclose(f);
	return ok;
}

bool parse_size(const char* s, int& w, int& h) { return std::sscanf(s, "%dx%d", &w, &h) == 2 && w > 0 && h > 0; }

} // namespace

int main(int argc, char** argv) {
	std::string plugin = "PT_HipWavefront", scene, out, engine, shading, resume, checkpoint;
	int width = 960, height = 540;                           // the reference's window, main.cpp:5
	int frames = 64, sppPerFrame = 1, depth = 0, moveAt = -1, resizeAt = -1, newW = 0, newH = 0, reportEvery = 0, snapshotEvery = 0;
	CameraData* cameraData = new CameraData();               // Program.cpp:36
	for (int a = 1; a < argc; a++) {
		auto need = [&](int n) { if (a + n >= argc) { std::fprintf(stderr, "%s needs %d argument(s)\n", argv[a], n); std::exit(2); } };
		if (!std::strcmp(argv[a], "--plugin")) { need(1); plugin = argv[++a]; }
		else if (!std::strcmp(argv[a], "--size")) { need(1); if (!parse_size(argv[++a], width, height)) { std::fprintf(stderr, "bad --size\n"); return 2; } }
		else if (!std::strcmp(argv[a], "--frames")) { need(1); frames = std::atoi(argv[++a]); }
		else if (!std::strcmp(argv[a], "--spp-per-frame")) { need(1); sppPerFrame = std::atoi(argv[++a]); }
		else if (!std::strcmp(argv[a], "--scene")) { need(1); scene = argv[++a]; }
		else if (!std::strcmp(argv[a], "--depth")) { need(1); depth = std::atoi(argv[++a]); }
		else if (!std::strcmp(argv[a], "--engine")) { need(1); engine = argv[++a]; }
		else if (!std::strcmp(argv[a], "--shading")) { need(1); shading = argv[++a]; }
		else if (!std::strcmp(argv[a], "--fov")) { need(1); cameraData->fieldOfView = (float)std::atof(argv[++a]); }
		else if (!std::strcmp(argv[a], "--pos")) { need(1); float x, y, z; if (std::sscanf(argv[++a], "%f,%f,%f", &x, &y, &z) != 3) { std::fprintf(stderr, "bad --pos\n"); return 2; } cameraData->position.x = x; cameraData->position.y = y; cameraData->position.z = z; }
		else if (!std::strcmp(argv[a], "--move-at")) { need(1); moveAt = std::atoi(argv[++a]); }
		else if (!std::strcmp(argv[a], "--resize-at")) { need(2); resizeAt = std::atoi(argv[++a]); if (!parse_size(argv[++a], newW, newH)) { std::fprintf(stderr, "bad --resize-at\n"); return 2; } }
		else if (!std::strcmp(argv[a], "--report-every")) { need(1); reportEvery = std::atoi(argv[++a]); }
		else if (!std::strcmp(argv[a], "--out")) { need(1); out = argv[++a]; }
		else if (!std::strcmp(argv[a], "--snapshot-every")) { need(1); snapshotEvery = std::atoi(argv[++a]); }   // with --out: <out>.<frame>.pfm every N frames (a progressive display's periodic readback)
		else if (!std::strcmp(argv[a], "--resume")) { need(1); resume = argv[++a]; }               // pt_load_state before the first frame
		else if (!std::strcmp(argv[a], "--checkpoint")) { need(1); checkpoint = argv[++a]; }       // pt_save_state after the last frame
		else { std::fprintf(stderr, "unknown argument %s\n", argv[a]); return 2; }
	}

	CreatePathTracerFunc CreatePathTracer = nullptr;
	DestroyPathTracerFunc DestroyPathTracer = nullptr;
	if (!LoadPathTracerLibrary(plugin.c_str(), &CreatePathTracer, &DestroyPathTracer)) return 3;
	void* lib = dlopen(plugin.c_str(), RTLD_NOW | RTLD_NOLOAD);
	if (!lib) lib = dlopen(("lib" + plugin + ".so").c_str(), RTLD_NOW | RTLD_NOLOAD);

	IPathTracer* pathTracer = CreatePathTracer(0, width, height, cameraData);
	if (!pathTracer) { std::fprintf(stderr, "Create returned null\n"); return 4; }

	auto last_error = sym<const char* (*)()>(lib, "pt_last_error");
	auto get_options = sym<int (*)(void*, pt_options*)>(lib, "pt_get_options");
	auto set_options = sym<int (*)(void*, const pt_options*)>(lib, "pt_set_options");
	auto load_scene = sym<int (*)(void*, const char*)>(lib, "pt_load_scene");
	auto get_stats = sym<int (*)(void*, pt_stats*)>(lib, "pt_get_stats");
	auto readback = sym<int (*)(void*, float*, size_t)>(lib, "pt_readback");
	auto fail = [&](const char* what) { std::fprintf(stderr, "%s failed: %s\n", what, last_error ? last_error() : "?"); return 5; };

	if (!scene.empty()) { if (!load_scene) return fail("pt_load_scene (not exported)"); if (load_scene(pathTracer, scene.c_str())) return fail("pt_load_scene"); }
	if (get_options && set_options) {
		pt_options o;
		if (get_options(pathTracer, &o)) return fail("pt_get_options");
		if (depth > 0) o.depth = (uint32_t)depth;
		if (sppPerFrame > 0) o.samples_per_begin = (uint32_t)sppPerFrame;
		if (engine == "wavefront") o.engine = PT_ENGINE_WAVEFRONT; else if (engine == "fused") o.engine = PT_ENGINE_FUSED;
		if (shading == "extended") o.shading = PT_SHADING_EXTENDED; else if (shading == "reference") o.shading = PT_SHADING_REFERENCE;
		if (set_options(pathTracer, &o)) return fail("pt_set_options");
	}

	if (!resume.empty()) {
		// the file brings options, camera, sample count, streams and mean; the host's camera follows it, or the first Update would restart
		auto load_state = sym<int (*)(void*, const char*)>(lib, "pt_load_state");
		auto get_camera = sym<int (*)(void*, pt_camera_data*)>(lib, "pt_get_camera");
		if (!load_state || !get_camera) return fail("pt_load_state (not exported)");
		if (load_state(pathTracer, resume.c_str())) return fail("pt_load_state");
		static_assert(sizeof(pt_camera_data) == sizeof(CameraData), "pt_camera_data is CameraData");
		if (get_camera(pathTracer, reinterpret_cast<pt_camera_data*>(cameraData))) return fail("pt_get_camera");
	}

	auto t0 = std::chrono::steady_clock::now(), tReport = t0;
	int framesSinceReport = 0;
	for (int frame = 0; frame < frames; frame++) {
		if (frame == moveAt) cameraData->fieldOfView -= 5.0f;                                        // Program.cpp:293-295
		if (frame == resizeAt) { width = newW; height = newH; pathTracer->Resize(width, height); }   // Program.cpp:208-209
		pathTracer->BeginDrawing();            // Program.cpp:57
		pathTracer->Update(cameraData);        // Program.cpp:58
		pathTracer->FinalizeDrawing();         // Program.cpp:70
		framesSinceReport++;
		if (snapshotEvery > 0 && !out.empty() && readback && (frame + 1) % snapshotEvery == 0 && frame + 1 < frames) {
			std::vector<float> rgba((size_t)width * height * 4);
			std::string name = out + "." + std::to_string(frame + 1) + ".pfm";
			if (readback(pathTracer, rgba.data(), rgba.size())) return fail("pt_readback");
			if (!write_pfm(name.c_str(), rgba, width, height)) { std::fprintf(stderr, "cannot write %s\n", name.c_str()); return 6; }
			std::printf("wrote %s\n", name.c_str());
		}
		if (reportEvery > 0 && framesSinceReport == reportEvery) {
			auto now = std::chrono::steady_clock::now();
			double ms = std::chrono::duration<double, std::milli>(now - tReport).count() / framesSinceReport;
			std::printf("frame %d: %.3f ms / frame, %.1f fps\n", frame + 1, ms, 1000.0 / ms);      // the window title of Program.cpp:78-90
			tReport = now; framesSinceReport = 0;
		}
	}
	double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	pt_stats st{};
	if (get_stats && get_stats(pathTracer, &st) == 0)
		std::printf("%d frames of %dx%d in %.2f ms: %.3f ms / frame, %.1f fps; %llu samples per pixel accumulated, %llu rays since the last restart, %.1f Mrays/s (device)\n",
		            frames, width, height, total_ms, total_ms / frames, 1000.0 * frames / total_ms, (unsigned long long)st.frames,
		            (unsigned long long)st.rays, st.total_render_ms > 0 ? st.rays / st.total_render_ms / 1e3 : 0.0);
	else
		std::printf("%d frames of %dx%d in %.2f ms: %.3f ms / frame, %.1f fps\n", frames, width, height, total_ms, total_ms / frames, 1000.0 * frames / total_ms);

	if (!checkpoint.empty()) {
		auto save_state = sym<int (*)(void*, const char*)>(lib, "pt_save_state");
		if (!save_state) return fail("pt_save_state (not exported)");
		if (save_state(pathTracer, checkpoint.c_str())) return fail("pt_save_state");
		std::printf("wrote %s\n", checkpoint.c_str());
	}
	if (!out.empty()) {
		if (!readback) return fail("pt_readback (not exported)");
		std::vector<float> rgba((size_t)width * height * 4);
		if (readback(pathTracer, rgba.data(), rgba.size())) return fail("pt_readback");
		if (!write_pfm(out.c_str(), rgba, width, height)) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 6; }
		std::printf("wrote %s\n", out.c_str());
	}
	DestroyPathTracer(pathTracer);          // Program.cpp:241
	delete cameraData;
	if (lib) dlclose(lib);
	UnloadPathTracerLibrary(plugin.c_str());
	return 0;
}

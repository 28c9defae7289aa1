// pt_plugin.cpp -- host side of the HIP plugin: class PathTracer : IPathTracer, the Create /
// Destroy exports and the additive pt_* C-ABI (include/pt_plugin.h).
//
// Plain C++17, no HIP headers: the device is reached only through the pt_hip_* C-ABI
// (include/pt_hip.h).  The state machine follows the reference's PT_CudaSimple/PathTracer.cu:
//   ctor  (:16-78)    pick device, allocate, seed RNG, upload the default scene
//   Update (:99-101)  restart flag = camera changed (overwritten every call)
//   BeginDrawing (:103-159)  restart if flagged, enqueue one frame, no host wait
//   FinalizeDrawing (:161-174)  no-op unless drawing; wait; frameNumber++
//   Resize (:176-184) reallocate per-pixel state, restart
//   dtor (:80-97)     finalize if mid-frame, free
// Differences, all forced by running headless: no GL texture (the running mean is a device
// buffer, see pt_readback), no cudaDeviceReset in the destructor (it would tear down other
// users of the device in the same process), errors are recorded instead of thrown.
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "IPathTracer.h"
#include "pt_hip.h"
#include "pt_plugin.h"
#include "pt_camera.hpp"
#include "pt_scene_io.hpp"

namespace {

thread_local std::string t_plugin_error;

int plugin_fail(int code, const std::string& msg) {
	t_plugin_error = msg;
	return code;
}
// forwards a pt_hip_* failure
int hip_layer_fail(int code, const char* where) {
	t_plugin_error = std::string(where) + ": " + pt_hip_last_error();
	return code;
}
#define PT_TRY(expr) do { int rc_ = (expr); if (rc_ != PT_OK) return hip_layer_fail(rc_, #expr); } while (0)

// Makes `device` the calling thread's current HIP device for the lifetime of the object and puts the previous one back
// afterwards.  The reference binds its device once (cudaSetDevice, PathTracer.cu:25) because it owns the process; a
// library that several instances -- one per GPU, each driven from any thread -- share must not rely on, nor change,
// what the caller has current: every method that allocates, copies or launches opens one of these first.
class DeviceScope {
	int _previous = -1;
	bool _switched = false;
public:
	explicit DeviceScope(int device) {
		if (device < 0) return;
		if (pt_hip_get_device(&_previous) != PT_OK) return;
		if (_previous != device) _switched = pt_hip_set_device(device) == PT_OK;
	}
	~DeviceScope() { if (_switched) pt_hip_set_device(_previous); }
	DeviceScope(const DeviceScope&) = delete;
	DeviceScope& operator=(const DeviceScope&) = delete;
};

// The reference's literal scene, PathTracer.cu:48-69.
void DefaultScene(std::vector<pt_vertex>& v, std::vector<pt_triangle>& t, std::vector<pt_material>& m) {
	const float V[8][4] = {
		{ -0.5f,  0.5f, 1.0f, 1.0f }, { 0.5f,  0.5f, 1.0f, 1.0f }, { 0.5f, -0.5f, 1.0f, 1.0f }, { -0.5f, -0.5f, 1.0f, 1.0f },
		{ -0.2f,  0.5f, 0.5f, 1.0f }, { 0.2f,  0.5f, 0.5f, 1.0f }, { 0.2f,  0.5f, 0.9f, 1.0f }, { -0.2f,  0.5f, 0.9f, 1.0f },
	};
	v.resize(8);
	for (int i = 0; i < 8; i++) v[i].position = pt_float4{ V[i][0], V[i][1], V[i][2], V[i][3] };
	t = { pt_triangle{ 0, 1, 2, 0 }, pt_triangle{ 2, 3, 0, 0 }, pt_triangle{ 4, 5, 6, 1 }, pt_triangle{ 6, 7, 4, 1 } };
	m.assign(2, pt_material{});
	m[0].type = PT_MAT_DIFFUSE;  m[0].color = pt_float4{ 1.0f, 0.2f, 0.2f, 1.0f };
	m[1].type = PT_MAT_EMISSIVE; m[1].color = pt_float4{ 10.0f, 10.0f, 10.0f, 10.0f };
}

class PathTracer final : public IPathTracer {
	int _width, _height;
	pt::Camera _camera;
	uint32_t _frameNumber = 0;
	bool _shouldRestartRendering = false;
	enum DrawState { Idle, Drawing } _drawState = Idle;
	uint32_t _inFlightSamples = 0;

	pt_options _opt{};
	uint32_t _pixelFirst = 0, _pixelCount = 0;

	struct DevicePtrs {
		uint32_t* rng = nullptr;        // per-pixel xorshift32 stream (RngState, Structures.cuh:18)
		void* mean = nullptr;           // float4 running mean: stands in for the GL surface
		void* scene = nullptr;          // packed Triangle/Vertex/Material blob
		uint64_t* stats = nullptr;
		void* workspace = nullptr;      // wavefront queues
	} _dev;
	size_t _workspaceBytes = 0;
	uint32_t _nTris = 0, _nMats = 0;
	bool _sceneHasBvh = false;          // recorded when the scene is packed (pt_hip_scene_has_bvh), passed with every frame
	int _device = -1;                   // resolved HIP ordinal of this instance, bound by every method (DeviceScope)
	void* _stream = nullptr;
	void* _evStart = nullptr;
	void* _evStop = nullptr;
	double _lastMs = 0.0, _totalMs = 0.0;
	uint64_t _renders = 0;
	uint32_t _frameFlags = 0;           // PT_FRAME_FAST_DIV once both image sizes are verified
	int _error = PT_OK;                 // sticky: once a device call fails the object stops rendering

public:
	PathTracer(int pixelWidth, int pixelHeight, const CameraData* cameraData)
		: _width(pixelWidth), _height(pixelHeight), _camera(cameraData, (float)pixelWidth / pixelHeight) {
		_opt.engine = PT_ENGINE_FUSED;
		_opt.rng_mode = PT_RNG_SEQUENTIAL;
		_opt.stride = PT_DEFAULT_STRIDE;
		_opt.depth = PT_DEFAULT_DEPTH;
		_opt.seed = PT_DEFAULT_SEED;
		_opt.samples_per_begin = 1;
		_opt.samples_per_pass = 16;
		_opt.device = -1;
	}

	void SetDevice(int device) { _opt.device = device; }

	int Initialize() {
		int count = 0;
		PT_TRY(pt_hip_device_count(&count));
		if (count <= 0) return plugin_fail(PT_ERR_HIP, "no HIP device visible");
		if (_opt.device >= count) return plugin_fail(PT_ERR_INVALID, "device ordinal beyond the visible HIP devices");
		if (_opt.device >= 0) _device = _opt.device;
		else PT_TRY(pt_hip_get_device(&_device));            // "whatever is current" is resolved once, here
		_opt.device = _device;
		DeviceScope bind(_device);
		std::cout << "Found " << count << " HIP device(s); path tracer running headless" << std::endl;
		PT_TRY(pt_hip_stream_create(&_stream));
		PT_TRY(pt_hip_event_create(&_evStart));
		PT_TRY(pt_hip_event_create(&_evStop));
		PT_TRY(pt_hip_malloc((void**)&_dev.stats, PT_STATS_WORDS * sizeof(uint64_t)));
		if (int rc = _AllocateDrawingMemory()) return rc;
		std::vector<pt_vertex> v; std::vector<pt_triangle> t; std::vector<pt_material> m;
		DefaultScene(v, t, m);
		return SetScene(v.data(), 8, t.data(), 4, m.data(), 2);      // seeds the streams too (_InitializeRendering)
	}

	~PathTracer() {
		std::cout << "Terminating HIP path tracer" << std::endl;
		DeviceScope bind(_device);
		if (_drawState == Drawing) FinalizeDrawing();
		if (_stream) pt_hip_stream_synchronize(_stream);
		pt_hip_free(_dev.rng); pt_hip_free(_dev.mean); pt_hip_free(_dev.scene);
		pt_hip_free(_dev.stats); pt_hip_free(_dev.workspace);
		pt_hip_event_destroy(_evStart); pt_hip_event_destroy(_evStop);
		pt_hip_stream_destroy(_stream);
	}

	// ---- IPathTracer ----
	void Update(const CameraData* cameraData) override {
		_shouldRestartRendering = _camera.TryUpdate(cameraData);
	}

	void BeginDrawing() override {
		DeviceScope bind(_device);
		_drawState = Drawing;
		_inFlightSamples = 0;
		if (_error) return;
		if (_shouldRestartRendering) {
			if ((_error = _InitializeRendering())) return;
		}
		_error = _Enqueue(_opt.samples_per_begin);
	}

	void FinalizeDrawing() override {
		if (_drawState != Drawing) return;
		DeviceScope bind(_device);
		_drawState = Idle;
		if (_error) return;
		float ms = 0.0f;
		if (_inFlightSamples) {
			if (pt_hip_event_elapsed_ms(_evStart, _evStop, &ms) != PT_OK) { _error = hip_layer_fail(PT_ERR_HIP, "FinalizeDrawing"); return; }
			_lastMs = ms; _totalMs += ms; _renders++;
		}
		if (pt_hip_stream_synchronize(_stream) != PT_OK) { _error = hip_layer_fail(PT_ERR_HIP, "FinalizeDrawing"); return; }
		_frameNumber += _inFlightSamples;
		_inFlightSamples = 0;
	}

	void Resize(const int pixelWidth, const int pixelHeight) override {
		if (pixelWidth <= 0 || pixelHeight <= 0) { _error = plugin_fail(PT_ERR_INVALID, "Resize: width and height must be positive"); return; }
		DeviceScope bind(_device);
		if (_drawState == Drawing) FinalizeDrawing();
		_width = pixelWidth;
		_height = pixelHeight;
		_camera.Resize(pixelWidth, pixelHeight);
		_opt.pixel_first = 0; _opt.pixel_count = 0;     // a tile of the old image means nothing now
		if ((_error = _AllocateDrawingMemory())) return;
		_error = _InitializeRendering();
	}

	// ---- additive API ----
	int Error() const { return _error; }

	int Render(uint32_t spp) {
		if (_error) return _error;
		DeviceScope bind(_device);
		if (_drawState == Drawing) FinalizeDrawing();
		_drawState = Drawing;
		_inFlightSamples = 0;
		if (_shouldRestartRendering) {
			if ((_error = _InitializeRendering())) return _error;
		}
		if ((_error = _Enqueue(spp))) return _error;
		FinalizeDrawing();
		return _error;
	}

	const pt_options& Options() const { return _opt; }

	int SetOptions(const pt_options& in) {
		if (_error) return _error;
		if (in.engine != PT_ENGINE_WAVEFRONT && in.engine != PT_ENGINE_FUSED) return plugin_fail(PT_ERR_INVALID, "unknown engine");
		if (in.rng_mode != PT_RNG_SEQUENTIAL && in.rng_mode != PT_RNG_STRIDED) return plugin_fail(PT_ERR_INVALID, "unknown rng_mode");
		if (in.depth < 1 || in.depth > PT_MAX_DEPTH) return plugin_fail(PT_ERR_INVALID, "depth must be 1..32");
		if (in.rng_mode == PT_RNG_STRIDED && in.stride == 0) return plugin_fail(PT_ERR_INVALID, "stride must be positive");
		if (in.samples_per_begin == 0) return plugin_fail(PT_ERR_INVALID, "samples_per_begin must be positive");
		if (in.shading != PT_SHADING_REFERENCE && in.shading != PT_SHADING_EXTENDED) return plugin_fail(PT_ERR_INVALID, "unknown shading");
		uint64_t n = (uint64_t)_width * _height;
		uint64_t cnt = in.pixel_count ? in.pixel_count : n - std::min<uint64_t>(in.pixel_first, n);
		if ((uint64_t)in.pixel_first + cnt > n || cnt == 0) return plugin_fail(PT_ERR_INVALID, "pixel range exceeds the image");
		if (in.device != _opt.device && in.device >= 0) return plugin_fail(PT_ERR_STATE, "device is chosen before Create (pt_set_default_device)");
		DeviceScope bind(_device);
		if (_drawState == Drawing) FinalizeDrawing();
		bool realloc = in.pixel_first != _opt.pixel_first || in.pixel_count != _opt.pixel_count ||
		               in.samples_per_pass != _opt.samples_per_pass || in.engine != _opt.engine || in.rng_mode != _opt.rng_mode;
		int dev = _opt.device;
		_opt = in;
		_opt.device = dev;
		if (_opt.samples_per_pass == 0) _opt.samples_per_pass = 1;
		if (realloc) { if ((_error = _AllocateDrawingMemory())) return _error; }
		return _error = _InitializeRendering();
	}

	// Restarts accumulation right away, like SetOptions and Resize (PathTracer.cu:176-184): a flag would be overwritten by
	// the next Update(same camera) (PathTracer.cu:99-101) and the new scene's samples folded into the old scene's mean.
	int SetScene(const pt_vertex* v, uint32_t nv, const pt_triangle* t, uint32_t nt, const pt_material* m, uint32_t nm) {
		if (_error) return _error;
		DeviceScope bind(_device);
		if (_drawState == Drawing) FinalizeDrawing();
		size_t bytes = pt_hip_scene_blob_bytes(nt, nm);
		std::vector<unsigned char> blob(bytes);
		PT_TRY(pt_hip_scene_pack(v, nv, t, nt, m, nm, blob.data(), bytes));
		int has_bvh = pt_hip_scene_has_bvh(blob.data(), bytes);
		if (has_bvh < 0) return hip_layer_fail(has_bvh, "pt_hip_scene_has_bvh");
		void* fresh = nullptr;
		PT_TRY(pt_hip_malloc(&fresh, bytes));
		if (int rc = pt_hip_memcpy_h2d(fresh, blob.data(), bytes, _stream)) { pt_hip_free(fresh); return hip_layer_fail(rc, "scene upload"); }
		PT_TRY(pt_hip_stream_synchronize(_stream));
		pt_hip_free(_dev.scene);
		_dev.scene = fresh; _nTris = nt; _nMats = nm; _sceneHasBvh = has_bvh != 0;
		return _error = _InitializeRendering();
	}

	int Readback(float* rgba, size_t n_floats) {
		if (_error) return _error;
		DeviceScope bind(_device);
		if (!rgba) return plugin_fail(PT_ERR_INVALID, "readback: null pointer");
		if (n_floats != (size_t)_pixelCount * 4) return plugin_fail(PT_ERR_INVALID, "readback: n_floats must be 4 * pixel_count");
		if (_drawState == Drawing) FinalizeDrawing();
		PT_TRY(pt_hip_memcpy_d2h(rgba, _dev.mean, n_floats * sizeof(float), _stream));
		return PT_OK;
	}
	int ReadbackRng(uint32_t* states, size_t n) {
		if (_error) return _error;
		DeviceScope bind(_device);
		if (!states) return plugin_fail(PT_ERR_INVALID, "readback_rng: null pointer");
		if (n != _pixelCount) return plugin_fail(PT_ERR_INVALID, "readback_rng: n must be pixel_count");
		if (_drawState == Drawing) FinalizeDrawing();
		PT_TRY(pt_hip_memcpy_d2h(states, _dev.rng, n * sizeof(uint32_t), _stream));
		return PT_OK;
	}
	int CopyMean(void* device_dst, size_t n_floats) {
		if (_error) return _error;
		DeviceScope bind(_device);
		if (!device_dst) return plugin_fail(PT_ERR_INVALID, "copy_mean: null pointer");
		if (n_floats != (size_t)_pixelCount * 4) return plugin_fail(PT_ERR_INVALID, "copy_mean: n_floats must be 4 * pixel_count");
		if (_drawState == Drawing) FinalizeDrawing();
		PT_TRY(pt_hip_memcpy_d2d(device_dst, _dev.mean, n_floats * sizeof(float), _stream));
		PT_TRY(pt_hip_stream_synchronize(_stream));
		return PT_OK;
	}
	int CopyMeanAsync(void* device_dst, size_t n_floats) {
		if (_error) return _error;
		DeviceScope bind(_device);
		if (!device_dst) return plugin_fail(PT_ERR_INVALID, "copy_mean_async: null pointer");
		if (n_floats != (size_t)_pixelCount * 4) return plugin_fail(PT_ERR_INVALID, "copy_mean_async: n_floats must be 4 * pixel_count");
		PT_TRY(pt_hip_memcpy_d2d(device_dst, _dev.mean, n_floats * sizeof(float), _stream));
		return PT_OK;
	}
	void* Stream() const { return _stream; }
	void* DeviceMean() { if (_drawState == Drawing) FinalizeDrawing(); return _dev.mean; }
	int Device() const { return _device; }

	int Stats(pt_stats* out) {
		if (_error) return _error;
		DeviceScope bind(_device);
		if (_drawState == Drawing) FinalizeDrawing();
		uint64_t words[PT_STATS_WORDS];
		PT_TRY(pt_hip_memcpy_d2h(words, _dev.stats, sizeof(words), _stream));
		std::memset(out, 0, sizeof(*out));
		out->rays = words[0]; out->diffuse_bounces = words[1]; out->paths = words[2];
		out->frames = _frameNumber;
		for (int b = 0; b < PT_MAX_DEPTH; b++) out->rays_per_bounce[b] = words[4 + b];
		out->last_render_ms = _lastMs; out->total_render_ms = _totalMs; out->renders = _renders;
		return PT_OK;
	}
	void Size(int* w, int* h, uint32_t* first, uint32_t* count) const {
		if (w) *w = _width;
		if (h) *h = _height;
		if (first) *first = _pixelFirst;
		if (count) *count = _pixelCount;
	}

private:
	pt_frame _Frame() const {
		pt_frame f{};
		f.width = _width; f.height = _height;
		f.pixel_first = _pixelFirst; f.pixel_count = _pixelCount;
		f.depth = _opt.depth; f.seed = _opt.seed; f.rng_mode = _opt.rng_mode; f.stride = _opt.stride;
		f.shading = _opt.shading;
		f.flags = _frameFlags | (_sceneHasBvh ? PT_FRAME_SCENE_BVH : 0u);
		f.origin = _camera.GetPosition();
		f.top_left = _camera.GetTopLeft();
		f.bottom_left = _camera.GetBottomLeft();
		f.bottom_right = _camera.GetBottomRight();
		return f;
	}

	// PathTracer.cu:217-233
	int _AllocateDrawingMemory() {
		uint64_t n = (uint64_t)_width * _height;
		if (n > 0x7FFFFFFFull) return plugin_fail(PT_ERR_LIMIT, "image larger than 2^31-1 pixels");
		_pixelFirst = _opt.pixel_first;
		_pixelCount = _opt.pixel_count ? _opt.pixel_count : (uint32_t)(n - _pixelFirst);
		pt_hip_free(_dev.rng); pt_hip_free(_dev.mean); pt_hip_free(_dev.workspace);
		_dev.rng = nullptr; _dev.mean = nullptr; _dev.workspace = nullptr; _workspaceBytes = 0;
		// the short division of the ray set-up is used only for sizes it is verified for (exhaustively, on the device)
		int okW = 0, okH = 0;
		PT_TRY(pt_hip_verify_fast_div((uint32_t)_width, &okW, _stream));
		PT_TRY(pt_hip_verify_fast_div((uint32_t)_height, &okH, _stream));
		_frameFlags = (okW && okH) ? PT_FRAME_FAST_DIV : 0u;
		PT_TRY(pt_hip_malloc((void**)&_dev.rng, (size_t)_pixelCount * sizeof(uint32_t)));
		PT_TRY(pt_hip_malloc(&_dev.mean, (size_t)_pixelCount * 4 * sizeof(float)));
		// the reference's surface starts white (Surface.cpp:99-115); frame 0 multiplies it by 0
		PT_TRY(pt_hip_fill_f32((float*)_dev.mean, 1.0f, (size_t)_pixelCount * 4, _stream));
		if (_opt.engine == PT_ENGINE_WAVEFRONT) {
			uint32_t spp = _opt.rng_mode == PT_RNG_STRIDED ? _opt.samples_per_pass : 1;
			_workspaceBytes = pt_hip_wavefront_workspace_bytes(_pixelCount, spp, _opt.depth);
			PT_TRY(pt_hip_malloc(&_dev.workspace, _workspaceBytes));
		}
		return PT_OK;
	}

	// PathTracer.cu:235-244
	int _InitializeRendering() {
		_shouldRestartRendering = false;
		_frameNumber = 0;
		_totalMs = 0.0; _renders = 0;
		pt_frame f = _Frame();
		PT_TRY(pt_hip_rng_seed(&f, _dev.rng, _stream));
		PT_TRY(pt_hip_memset(_dev.stats, 0, PT_STATS_WORDS * sizeof(uint64_t), _stream));
		return PT_OK;
	}

	// PathTracer.cu:116-157, for `samples` samples
	int _Enqueue(uint32_t samples) {
		if (samples == 0) return PT_OK;
		if ((uint64_t)_frameNumber + samples > 0xFFFFFFFFull) return plugin_fail(PT_ERR_LIMIT, "frame counter would overflow");
		pt_frame f = _Frame();
		uint32_t frame_first = _opt.sample_offset + _frameNumber;
		PT_TRY(pt_hip_event_record(_evStart, _stream));
		if (_opt.engine == PT_ENGINE_WAVEFRONT) {
			PT_TRY(pt_hip_render_wavefront(&f, _dev.scene, _nTris, _nMats, frame_first, _frameNumber, samples, _opt.samples_per_pass,
			                               _dev.rng, _dev.mean, _dev.workspace, _workspaceBytes, _dev.stats, _stream));
		} else {
			PT_TRY(pt_hip_render_fused(&f, _dev.scene, _nTris, _nMats, frame_first, _frameNumber, samples,
			                           _dev.rng, _dev.mean, _dev.stats, _stream));
		}
		PT_TRY(pt_hip_event_record(_evStop, _stream));
		_inFlightSamples = samples;
		return PT_OK;
	}
};

int g_default_device = -1;

PathTracer* as_tracer(pt_handle* h) { return static_cast<PathTracer*>(static_cast<IPathTracer*>(h)); }

} // namespace

// ---- the reference's two exports (PathTracer.cuh:69-76) ----
extern "C" IPathTracer* Create(const unsigned int glTexture, const int pixelWidth, const int pixelHeight, const CameraData* cameraData) {
	(void)glTexture;   // GL texture of the reference's display surface: no display path here
	if (pixelWidth <= 0 || pixelHeight <= 0 || !cameraData) {
		plugin_fail(PT_ERR_INVALID, "Create: width/height must be positive and cameraData non-null");
		std::cout << "Failed to create HIP path tracer: " << t_plugin_error << std::endl;
		return nullptr;
	}
	PathTracer* tracer = new PathTracer(pixelWidth, pixelHeight, cameraData);
	if (g_default_device >= 0) tracer->SetDevice(g_default_device);   // applied inside Initialize
	if (tracer->Initialize() != PT_OK) {
		std::string msg = t_plugin_error;
		std::cout << "Failed to create HIP path tracer: " << msg << std::endl;
		delete tracer;
		t_plugin_error = msg;
		return nullptr;
	}
	return tracer;
}

extern "C" void Destroy(IPathTracer* pathTracer) {
	delete static_cast<PathTracer*>(pathTracer);
}

// ---- additive C-ABI ----
extern "C" {

const char* pt_last_error(void) { return t_plugin_error.c_str(); }

int pt_set_default_device(int device) { g_default_device = device; return PT_OK; }

#define PT_HANDLE(h) if (!(h)) return plugin_fail(PT_ERR_INVALID, "null handle"); PathTracer* tracer = as_tracer(h)

int pt_update(pt_handle* h, const pt_camera_data* camera) {
	PT_HANDLE(h);
	if (!camera) return plugin_fail(PT_ERR_INVALID, "camera is null");
	static_assert(sizeof(pt_camera_data) == sizeof(CameraData), "pt_camera_data must mirror CameraData");
	CameraData c;
	std::memcpy(static_cast<void*>(&c), camera, sizeof(c));
	static_cast<IPathTracer*>(tracer)->Update(&c);
	return tracer->Error();
}
int pt_begin_drawing(pt_handle* h) { PT_HANDLE(h); static_cast<IPathTracer*>(tracer)->BeginDrawing(); return tracer->Error(); }
int pt_finalize_drawing(pt_handle* h) { PT_HANDLE(h); static_cast<IPathTracer*>(tracer)->FinalizeDrawing(); return tracer->Error(); }
int pt_resize(pt_handle* h, int w, int hgt) { PT_HANDLE(h); static_cast<IPathTracer*>(tracer)->Resize(w, hgt); return tracer->Error(); }

int pt_get_options(pt_handle* h, pt_options* out) {
	PT_HANDLE(h);
	if (!out) return plugin_fail(PT_ERR_INVALID, "out is null");
	*out = tracer->Options();
	return PT_OK;
}
int pt_set_options(pt_handle* h, const pt_options* in) {
	PT_HANDLE(h);
	if (!in) return plugin_fail(PT_ERR_INVALID, "options is null");
	return tracer->SetOptions(*in);
}
int pt_set_scene(pt_handle* h, const pt_vertex* v, uint32_t nv, const pt_triangle* t, uint32_t nt, const pt_material* m, uint32_t nm) {
	PT_HANDLE(h);
	return tracer->SetScene(v, nv, t, nt, m, nm);
}
int pt_scene_file_read(const char* path, pt_vertex** v, uint32_t* nv, pt_triangle** t, uint32_t* nt, pt_material** m, uint32_t* nm) {
	if (!path || !v || !nv || !t || !nt || !m || !nm) return plugin_fail(PT_ERR_INVALID, "scene_file_read: null pointer");
	ptio::Scene sc;
	std::string err = ptio::read(path, sc);
	if (!err.empty()) return plugin_fail(PT_ERR_INVALID, err);
	auto dup = [](const void* src, size_t bytes) { void* p = std::malloc(bytes ? bytes : 1); if (p && bytes) std::memcpy(p, src, bytes); return p; };
	*v = (pt_vertex*)dup(sc.vertices.data(), sc.vertices.size() * sizeof(pt_vertex));
	*t = (pt_triangle*)dup(sc.triangles.data(), sc.triangles.size() * sizeof(pt_triangle));
	*m = (pt_material*)dup(sc.materials.data(), sc.materials.size() * sizeof(pt_material));
	if (!*v || !*t || !*m) { std::free(*v); std::free(*t); std::free(*m); return plugin_fail(PT_ERR_NOMEM, "scene_file_read: out of memory"); }
	*nv = (uint32_t)sc.vertices.size(); *nt = (uint32_t)sc.triangles.size(); *nm = (uint32_t)sc.materials.size();
	return PT_OK;
}
void pt_scene_file_free(void* array) { std::free(array); }
int pt_scene_file_write(const char* path, const pt_vertex* v, uint32_t nv, const pt_triangle* t, uint32_t nt, const pt_material* m, uint32_t nm) {
	if (!path || (!v && nv) || (!t && nt) || (!m && nm)) return plugin_fail(PT_ERR_INVALID, "scene_file_write: null pointer");
	std::string err = ptio::write(path, v, nv, t, nt, m, nm);
	return err.empty() ? PT_OK : plugin_fail(PT_ERR_INVALID, err);
}
int pt_load_scene(pt_handle* h, const char* path) {
	PT_HANDLE(h);
	if (!path) return plugin_fail(PT_ERR_INVALID, "load_scene: null path");
	ptio::Scene sc;
	std::string err = ptio::read(path, sc);
	if (!err.empty()) return plugin_fail(PT_ERR_INVALID, err);
	return tracer->SetScene(sc.vertices.data(), (uint32_t)sc.vertices.size(), sc.triangles.data(), (uint32_t)sc.triangles.size(),
	                        sc.materials.data(), (uint32_t)sc.materials.size());
}
int pt_render(pt_handle* h, uint32_t spp) { PT_HANDLE(h); return tracer->Render(spp); }
int pt_readback(pt_handle* h, float* rgba, size_t n_floats) { PT_HANDLE(h); return tracer->Readback(rgba, n_floats); }
int pt_readback_rng(pt_handle* h, uint32_t* states, size_t n) { PT_HANDLE(h); return tracer->ReadbackRng(states, n); }
int pt_copy_mean(pt_handle* h, void* device_dst, size_t n_floats) { PT_HANDLE(h); return tracer->CopyMean(device_dst, n_floats); }
int pt_copy_mean_async(pt_handle* h, void* device_dst, size_t n_floats) { PT_HANDLE(h); return tracer->CopyMeanAsync(device_dst, n_floats); }
void* pt_stream(pt_handle* h) { return h ? as_tracer(h)->Stream() : nullptr; }
void* pt_device_mean(pt_handle* h) { return h ? as_tracer(h)->DeviceMean() : nullptr; }
int pt_get_stats(pt_handle* h, pt_stats* out) {
	PT_HANDLE(h);
	if (!out) return plugin_fail(PT_ERR_INVALID, "out is null");
	return tracer->Stats(out);
}
int pt_get_size(pt_handle* h, int* w, int* hgt, uint32_t* first, uint32_t* count) { PT_HANDLE(h); tracer->Size(w, hgt, first, count); return PT_OK; }

} // extern "C"

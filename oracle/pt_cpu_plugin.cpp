// pt_cpu_plugin.cpp -- the CPU ORACLE behind the reference's plugin interface
// (TEST INFRASTRUCTURE, NOT PRODUCT CODE; see oracle/pt_oracle.h).
//
// libPT_CpuOracle.so exports Create / Destroy (reference PathTracer.cuh:69-76) and returns an
// IPathTracer whose frames are rendered by oracle/pt_oracle.c on the host cores.  It is config
// C1 of BASELINE.json ("reference default scene via IPathTracer, headless, 640x480, 1 spp, on the
// host CPU path") and the cpu_baseline leg of bench.py.  State machine as in the reference's
// PathTracer.cu:99-184; rendering happens synchronously inside BeginDrawing.
#include <cstring>
#include <iostream>
#include <vector>

#include "IPathTracer.h"
#include "pt_oracle.h"

namespace {

class CpuPathTracer final : public IPathTracer {
	int _width, _height;
	CameraData _camera;
	uint32_t _frameNumber = 0;
	bool _shouldRestartRendering = false;
	bool _drawing = false;
	std::vector<uint32_t> _rng;
	std::vector<pto_f4> _mean;
	pto_vertex _v[8]; pto_triangle _t[4]; pto_material _m[2];

public:
	CpuPathTracer(int w, int h, const CameraData* cam) : _width(w), _height(h), _camera(*cam) {
		pto_default_scene(_v, _t, _m);
		_Allocate();
		_InitializeRendering();
	}
	void Update(const CameraData* cam) override {
		if (_camera != *cam) { _camera = *cam; _shouldRestartRendering = true; }
		else _shouldRestartRendering = false;                       // PathTracer.cu:99-101
	}
	void BeginDrawing() override {
		_drawing = true;
		if (_shouldRestartRendering) _InitializeRendering();
		pto_camera cam;
		std::memcpy(&cam, static_cast<const void*>(&_camera), sizeof(cam));
		pto_f4 c[4];
		pto_camera_corners(&cam, (float)_width / _height, c);
		pto_render_params p{};
		p.width = _width; p.height = _height; p.depth = 10; p.seed = 1337u;
		p.rng_mode = PTO_RNG_SEQUENTIAL; p.stride = 35; p.frame_first = _frameNumber; p.spp = 1;
		p.origin = c[0]; p.top_left = c[1]; p.bottom_left = c[2]; p.bottom_right = c[3];
		p.threads = 0;
		pto_render(&p, _v, 8, _t, 4, _m, 2, nullptr, 0, (uint32_t)_mean.size(), _rng.data(), _mean.data(), nullptr, nullptr);
	}
	void FinalizeDrawing() override {
		if (!_drawing) return;
		_drawing = false;
		_frameNumber++;
	}
	void Resize(const int w, const int h) override {
		_width = w; _height = h;
		_Allocate();
		_InitializeRendering();
	}
	int Readback(float* rgba, size_t n) const {
		if (n != _mean.size() * 4) return -1;
		std::memcpy(rgba, _mean.data(), n * sizeof(float));
		return 0;
	}

private:
	void _Allocate() {
		size_t n = (size_t)_width * _height;
		_rng.assign(n, 0u);
		_mean.assign(n, pto_f4{ 1.0f, 1.0f, 1.0f, 1.0f });
	}
	void _InitializeRendering() {
		_shouldRestartRendering = false;
		_frameNumber = 0;
		pto_rng_seed_pixels(1337u, nullptr, 0, (uint32_t)_rng.size(), _rng.data());
	}
};

} // namespace

static_assert(sizeof(pto_camera) == sizeof(CameraData), "pto_camera must mirror CameraData");

extern "C" IPathTracer* Create(const unsigned int, const int w, const int h, const CameraData* cam) {
	if (w <= 0 || h <= 0 || !cam) return nullptr;
	return new CpuPathTracer(w, h, cam);
}
extern "C" void Destroy(IPathTracer* p) { delete static_cast<CpuPathTracer*>(p); }
extern "C" int pt_readback(void* h, float* rgba, size_t n_floats) {
	return h ? static_cast<CpuPathTracer*>(static_cast<IPathTracer*>(h))->Readback(rgba, n_floats) : -1;
}

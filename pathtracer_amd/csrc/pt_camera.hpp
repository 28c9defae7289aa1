// pt_camera.hpp -- host-side camera of the plugin: CameraData + aspect ratio -> the four
// float4 kernel arguments of ray generation.  Mirrors the behaviour of the reference's
// PT_CudaSimple/Camera.h:25-83, including its two quirks:
//   * GLM_ROTATE_VECTOR(v, q) = q * v * conjugate(q) parses as (q*v) * conj(q), and GLM's
//     vec4 * quat is inverse(quat) * vec4, so the rotation is applied twice;
//   * corners are position + direction, and the kernel does not subtract the origin again.
// GLM is not a dependency: the handful of fp32 operations are written out in GLM's order
// (include/glm/detail/type_quat.inl:343-368, ext/quaternion_common.inl:113-122,
// detail/func_geometric.inl:74-77) so host results match a GLM build operation for operation.
#pragma once

#include <cmath>

#include "CameraData.h"
#include "pt_hip.h"

// build with -ffp-contract=off (csrc/Makefile): fp32 results must not depend on FMA fusion

namespace pt {

struct Q4 { float x, y, z, w; };
struct V3 { float x, y, z; };

inline V3 Cross(V3 a, V3 b) {
	return V3{ a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y };
}

// quat * vec3
inline V3 Rotate(Q4 q, V3 v) {
	V3 u{ q.x, q.y, q.z };
	V3 uv = Cross(u, v);
	V3 uuv = Cross(u, uv);
	return V3{ v.x + ((uv.x * q.w) + uuv.x) * 2.0f, v.y + ((uv.y * q.w) + uuv.y) * 2.0f, v.z + ((uv.z * q.w) + uuv.z) * 2.0f };
}

// (q * v) * conjugate(q)  ==  inverse(conjugate(q)) * (q * v)
inline pt_float4 RotateVector(pt_float4 v, Q4 q) {
	V3 once = Rotate(q, V3{ v.x, v.y, v.z });
	Q4 c{ -q.x, -q.y, -q.z, q.w };                        // conjugate(q)
	float d = (c.w * c.w + c.x * c.x) + (c.y * c.y + c.z * c.z);
	Q4 inv{ -c.x / d, -c.y / d, -c.z / d, c.w / d };      // conjugate(c) / dot(c, c)
	V3 twice = Rotate(inv, once);
	return pt_float4{ twice.x, twice.y, twice.z, v.w };
}

class Camera {
	CameraData _data;
	float _aspectRatio;   // width / height (the reference's member comment says the opposite; its code is w/h)

public:
	Camera(const CameraData* cameraData, float aspectRatio) : _data(*cameraData), _aspectRatio(aspectRatio) {}

	// true when the camera changed (by value) -- the caller restarts accumulation
	bool TryUpdate(const CameraData* data) {
		if (_data != *data) { _data = *data; return true; }
		return false;
	}
	void Resize(int width, int height) { _aspectRatio = (float)width / height; }
	const CameraData& Data() const { return _data; }

	pt_float4 GetPosition() const { return pt_float4{ _data.position.x, _data.position.y, _data.position.z, _data.position.w }; }
	pt_float4 GetTopLeft() const { return Corner(+0.5f, -1.0f); }
	pt_float4 GetBottomLeft() const { return Corner(-0.5f, -1.0f); }
	pt_float4 GetBottomRight() const { return Corner(-0.5f, +1.0f); }

private:
	float ForwardZ() const {
		return 2.0f / std::tan((_data.fieldOfView * 0.5f) * static_cast<float>(0.01745329251994329576923690768489));
	}
	// forward +/- 0.5*up -/+ (0.5*aspect)*right, evaluated left to right as the reference writes it
	pt_float4 Corner(float upHalf, float rightSign) const {
		float right = 0.5f * _aspectRatio;
		pt_float4 local;
		float x0 = 0.0f + upHalf * 0.0f, y0 = 0.0f + upHalf * 1.0f, z0 = ForwardZ() + upHalf * 0.0f, w0 = 0.0f + upHalf * 0.0f;
		if (upHalf < 0.0f) {   // forward - 0.5f * up
			x0 = 0.0f - 0.5f * 0.0f; y0 = 0.0f - 0.5f * 1.0f; z0 = ForwardZ() - 0.5f * 0.0f; w0 = 0.0f - 0.5f * 0.0f;
		}
		if (rightSign < 0.0f) local = pt_float4{ x0 - right * 1.0f, y0 - right * 0.0f, z0 - right * 0.0f, w0 - right * 0.0f };
		else local = pt_float4{ x0 + right * 1.0f, y0 + right * 0.0f, z0 + right * 0.0f, w0 + right * 0.0f };
		Q4 q{ _data.rotation.x, _data.rotation.y, _data.rotation.z, _data.rotation.w };
		pt_float4 dir = RotateVector(local, q);
		return pt_float4{ _data.position.x + dir.x, _data.position.y + dir.y, _data.position.z + dir.z, _data.position.w + dir.w };
	}
};

} // namespace pt

// CameraData.h -- camera POD handed across the plugin boundary.
//
// Layout- and semantics-compatible with the reference's include/CameraData.h:14-26:
//   vec4 position (x,y,z,w) = (0,0,0,1); quat rotation stored x,y,z,w = identity;
//   float fieldOfView = 90 (vertical, degrees).  36 bytes, no padding.
// The reference spells the members glm::vec4 / glm::quat.  When GLM is on the include
// path we use it, so a host written against the reference compiles unchanged; otherwise
// a minimal layout-compatible stand-in lives in namespace glm (only storage, ==, and the
// two constructors the struct's defaults need).
#pragma once

#if defined(__has_include)
#  if __has_include("glm/glm.hpp") && __has_include("glm/ext.hpp")
#    define PT_CAMERADATA_HAVE_GLM 1
#  endif
#endif

#ifdef PT_CAMERADATA_HAVE_GLM
#  include "glm/glm.hpp"
#  include "glm/ext.hpp"
#else
namespace glm {
struct vec3 {
	float x, y, z;
	constexpr vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
struct vec4 {
	float x, y, z, w;
	constexpr vec4() : x(0.0f), y(0.0f), z(0.0f), w(0.0f) {}
	constexpr vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
	constexpr bool operator==(const vec4& o) const { return x == o.x && y == o.y && z == o.z && w == o.w; }
};
// Storage order x,y,z,w (GLM default, no GLM_FORCE_QUAT_DATA_WXYZ); constructor order w,x,y,z.
struct quat {
	float x, y, z, w;
	constexpr quat() : x(0.0f), y(0.0f), z(0.0f), w(1.0f) {}
	constexpr quat(float w_, float x_, float y_, float z_) : x(x_), y(y_), z(z_), w(w_) {}
	constexpr bool operator==(const quat& o) const { return x == o.x && y == o.y && z == o.z && w == o.w; }
};
} // namespace glm
#endif

struct CameraData {
	glm::vec4 position = glm::vec4(0.0f, 0.0f, 0.0f, 1.0f);
#ifdef PT_CAMERADATA_HAVE_GLM
	glm::quat rotation = glm::quat(glm::vec3(0.0f, 0.0f, 0.0f));
#else
	glm::quat rotation = glm::quat(1.0f, 0.0f, 0.0f, 0.0f);
#endif
	float fieldOfView = 90.0f; // vertical, degrees

	// Change detection is by value (reference Camera.h:37-43 relies on it).
	inline bool operator==(const CameraData& other) const {
		return position == other.position && rotation == other.rotation && fieldOfView == other.fieldOfView;
	}
	inline bool operator!=(const CameraData& other) const { return !(*this == other); }
};

static_assert(sizeof(CameraData) == 36, "CameraData must stay 36 bytes: vec4 + quat + float");

// IPathTracer.h -- the plugin interface every path-tracer backend implements.
//
// ABI-identical to the reference's include/IPathTracer.h:5-14: four pure virtuals in this
// exact order (vtable slots 0..3) and NO virtual destructor, so objects are released only
// through the library's Destroy export.  Call protocol per frame (reference
// PathTracer/Program.cpp:57-58,70): BeginDrawing() enqueues one sample per pixel
// asynchronously, Update() hands over the camera for the NEXT frame, FinalizeDrawing()
// blocks until the sample has been folded into the running mean.
#pragma once

#include "CameraData.h"

class IPathTracer {
public:
	// slot 0: camera for the next frame; a value change restarts accumulation (reference PathTracer.cu:99-108).
	virtual void Update(const CameraData* cameraData) = 0;
	// slot 1: start one sample per pixel; returns without waiting for the device (PathTracer.cu:103-159).
	virtual void BeginDrawing() = 0;
	// slot 2: wait for the sample started by BeginDrawing; no-op when nothing is in flight (PathTracer.cu:161-174).
	virtual void FinalizeDrawing() = 0;
	// slot 3: new image size; reallocates per-pixel state and restarts accumulation (PathTracer.cu:176-184).
	virtual void Resize(const int pixelWidth, const int pixelHeight) = 0;
};

// Factory exports, resolved by the names "Create" and "Destroy" (reference IPathTracerManagement.h:13-14).
// glTexture is the GL texture name of the reference's display surface; headless backends ignore it (pass 0).
typedef IPathTracer* (*CreatePathTracerFunc)(const unsigned int glTexture, const int pixelWidth, const int pixelHeight, const CameraData* cameraData);
typedef void (*DestroyPathTracerFunc)(IPathTracer*);

// IPathTracerManagement.h -- load / unload a path-tracer plugin by name (POSIX twin).
//
// Same two functions, signatures, bool convention and stdout message shape as the reference's
// Win32-only include/IPathTracerManagement.h:9-47 (LoadLibraryA/GetProcAddress/FreeLibrary),
// here over dlopen/dlsym/dlclose.  A bare name such as "PT_HipWavefront" is tried as given,
// then as "lib<name>.so" (the loader's search path applies, like the DLL search path did).
// The Windows numeric GetLastError() has no POSIX counterpart; dlerror()'s text is printed
// in its place.
#pragma once

#include "IPathTracer.h"

#include <dlfcn.h>
#include <iostream>
#include <map>
#include <string>

namespace pt_detail {
inline std::map<std::string, void*>& LoadedLibraries() {
	static std::map<std::string, void*> handles;
	return handles;
}
inline bool DllFail(const char* description, const char* libraryName) {
	const char* err = dlerror();
	std::cout << description << " (library '" << libraryName << "', error '" << (err ? err : "none") << "')" << std::endl;
	return false;
}
inline void* OpenByName(const char* libraryName) {
	void* handle = dlopen(libraryName, RTLD_NOW | RTLD_LOCAL);
	if (!handle) {
		std::string decorated = std::string("lib") + libraryName + ".so";
		handle = dlopen(decorated.c_str(), RTLD_NOW | RTLD_LOCAL);
	}
	return handle;
}
} // namespace pt_detail

inline bool LoadPathTracerLibrary(const char* libraryName, CreatePathTracerFunc* createFuncPtr, DestroyPathTracerFunc* destroyFuncPtr) {
	void* handle = pt_detail::OpenByName(libraryName);
	if (!handle) return pt_detail::DllFail("Unable to load library.", libraryName);

	CreatePathTracerFunc create = (CreatePathTracerFunc)dlsym(handle, "Create");
	DestroyPathTracerFunc destroy = (DestroyPathTracerFunc)dlsym(handle, "Destroy");
	if (!create || !destroy) {
		bool result = pt_detail::DllFail("Unable to get Create and Destroy function addresses.", libraryName);
		dlclose(handle);
		return result;
	}
	pt_detail::LoadedLibraries()[libraryName] = handle;
	*createFuncPtr = create;
	*destroyFuncPtr = destroy;
	return true;
}

inline bool UnloadPathTracerLibrary(const char* libraryName) {
	auto& handles = pt_detail::LoadedLibraries();
	auto it = handles.find(libraryName);
	if (it == handles.end()) return pt_detail::DllFail("Unable to get handle of module.", libraryName);
	if (dlclose(it->second) != 0) return pt_detail::DllFail("Unable to free library.", libraryName);
	handles.erase(it);
	return true;
}

/*
 * config.h for the reference-build probe.  The reference asks every user to
 * write this file for their machine (config.h.dist:1-6 in the reference);
 * this one serves oracle/ref_harness.cpp.  OBJ_PATH is never opened: the
 * probe fills the mesh through the Mesh/Material public members instead of
 * Mesh::loadMesh, which needs the Corona image library.
 *
 * imageWriter.h:165-167 calls std::powf, which MSVC's <cmath> declares and
 * libstdc++ 11 does not; the using-declaration below names the C library's
 * powf there (the same function std::pow(float, float) resolves to).
 */
#ifndef __CONFIG_H__
#define __CONFIG_H__
#include <cctype>
#include <cmath>
#include <cstring>
#include <limits>

namespace std {
using ::powf;
}

#define OBJ_PATH "unused.obj"

const unsigned WindowSize_X = 64;
const unsigned WindowSize_Y = 64;
const unsigned WindowPos_X = 0;
const unsigned WindowPos_Y = 0;
const unsigned Thread_Amount = 1;

#endif

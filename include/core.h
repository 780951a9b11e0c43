// core.h -- POD types of the Radiance host API (facade header, source-compatible with the
// reference's radiance/src/core.h so that reference callers compile unchanged).
//
// The reference takes its vector / matrix types from assimp (core.h:3-5,14-16).  When assimp's
// headers are on the include path they are used as-is; otherwise the few members the API and its
// callers rely on (x/y/z, operator[], +, *scalar, a 16-argument row-major matrix constructor,
// identity default, operator*, aiMatrix3x3::Rotation) are provided here with the same spelling.
#pragma once

#include <cmath>
#include <cstdint>
#include <list>
#include <vector>

#if defined(__has_include)
#if __has_include(<assimp/scene.h>)
#define RD_HAVE_ASSIMP 1
#endif
#endif

#ifdef RD_HAVE_ASSIMP
#include <assimp/Importer.hpp>
#include <assimp/postprocess.h>
#include <assimp/scene.h>
#else
typedef unsigned int uint;
template <typename T> struct aiVector2t { T x{}, y{}; aiVector2t() = default; aiVector2t(T a, T b) : x(a), y(b) {} };
template <typename T> struct aiVector3t {
    T x{}, y{}, z{};
    aiVector3t() = default;
    aiVector3t(T a, T b, T c) : x(a), y(b), z(c) {}
    T& operator[](unsigned i) { return (&x)[i]; }
    const T& operator[](unsigned i) const { return (&x)[i]; }
};
template <typename T> inline aiVector3t<T> operator+(const aiVector3t<T>& a, const aiVector3t<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> inline aiVector3t<T> operator-(const aiVector3t<T>& a, const aiVector3t<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> inline aiVector3t<T> operator*(const aiVector3t<T>& a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T> struct aiMatrix3x3t {
    T a1{1}, a2{0}, a3{0}, b1{0}, b2{1}, b3{0}, c1{0}, c2{0}, c3{1};
    // rotation by `a` radians about `axis` (x, y or z unit vectors are what callers pass)
    static aiMatrix3x3t& Rotation(T a, const aiVector3t<T>& axis, aiMatrix3x3t& out)
    {
        const T c = std::cos(a), s = std::sin(a), t = 1 - c, x = axis.x, y = axis.y, z = axis.z;
        out.a1 = t * x * x + c;     out.a2 = t * x * y - s * z; out.a3 = t * x * z + s * y;
        out.b1 = t * x * y + s * z; out.b2 = t * y * y + c;     out.b3 = t * y * z - s * x;
        out.c1 = t * x * z - s * y; out.c2 = t * y * z + s * x; out.c3 = t * z * z + c;
        return out;
    }
};
template <typename T> inline aiMatrix3x3t<T> operator*(const aiMatrix3x3t<T>& l, const aiMatrix3x3t<T>& m)
{
    aiMatrix3x3t<T> r;
    r.a1 = m.a1 * l.a1 + m.b1 * l.a2 + m.c1 * l.a3; r.a2 = m.a2 * l.a1 + m.b2 * l.a2 + m.c2 * l.a3; r.a3 = m.a3 * l.a1 + m.b3 * l.a2 + m.c3 * l.a3;
    r.b1 = m.a1 * l.b1 + m.b1 * l.b2 + m.c1 * l.b3; r.b2 = m.a2 * l.b1 + m.b2 * l.b2 + m.c2 * l.b3; r.b3 = m.a3 * l.b1 + m.b3 * l.b2 + m.c3 * l.b3;
    r.c1 = m.a1 * l.c1 + m.b1 * l.c2 + m.c1 * l.c3; r.c2 = m.a2 * l.c1 + m.b2 * l.c2 + m.c2 * l.c3; r.c3 = m.a3 * l.c1 + m.b3 * l.c2 + m.c3 * l.c3;
    return r;
}
template <typename T> inline aiVector3t<T> operator*(const aiMatrix3x3t<T>& m, const aiVector3t<T>& v)
{
    return {m.a1 * v.x + m.a2 * v.y + m.a3 * v.z, m.b1 * v.x + m.b2 * v.y + m.b3 * v.z, m.c1 * v.x + m.c2 * v.y + m.c3 * v.z};
}
template <typename T> struct aiMatrix4x4t {      // row-major: a = row 0 ... d = row 3
    T a1{1}, a2{0}, a3{0}, a4{0}, b1{0}, b2{1}, b3{0}, b4{0}, c1{0}, c2{0}, c3{1}, c4{0}, d1{0}, d2{0}, d3{0}, d4{1};
    aiMatrix4x4t() = default;
    aiMatrix4x4t(T _a1, T _a2, T _a3, T _a4, T _b1, T _b2, T _b3, T _b4, T _c1, T _c2, T _c3, T _c4, T _d1, T _d2, T _d3, T _d4)
        : a1(_a1), a2(_a2), a3(_a3), a4(_a4), b1(_b1), b2(_b2), b3(_b3), b4(_b4), c1(_c1), c2(_c2), c3(_c3), c4(_c4),
          d1(_d1), d2(_d2), d3(_d3), d4(_d4) {}
};
// term order of assimp's aiMatrix4x4t::operator*= (m = right operand): m.a1*a1 + m.b1*a2 + m.c1*a3 + m.d1*a4
template <typename T> inline aiMatrix4x4t<T> operator*(const aiMatrix4x4t<T>& l, const aiMatrix4x4t<T>& m)
{
    const T* L = &l.a1; const T* M = &m.a1;
    aiMatrix4x4t<T> r;
    T* R = &r.a1;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            R[4 * i + j] = M[j] * L[4 * i] + M[4 + j] * L[4 * i + 1] + M[8 + j] * L[4 * i + 2] + M[12 + j] * L[4 * i + 3];
    return r;
}
typedef aiVector3t<float> aiVector3f;
typedef aiVector3t<float> aiVector3D;
typedef aiMatrix3x3t<float> aiMatrix3x3;
typedef aiMatrix4x4t<float> aiMatrix4x4;
#endif // RD_HAVE_ASSIMP

namespace RD
{
typedef aiVector2t<float> Vec2;
typedef aiVector3f Vec3;
typedef aiMatrix4x4 Mat4x4;

// mesh triangle: three indices into Mesh::vertexData
struct Triangle { unsigned int idx0, idx1, idx2; };

// ---- acceleration-structure blob records (byte layout shared with the device) -------------------
struct AccelStructTop    { unsigned int type, nodeByteOffset, instByteOffset, totalBufferSize; };
struct AccelStructBottom { unsigned int type, nodeByteOffset, faceByteOffset, vertexOffset; };
struct DeviceInstance    { Mat4x4 transform; unsigned int SBTOffset, instanceID, customInstanceID, bottomAccelStructOffset; };
struct DeviceBVHNode {
    aiVector3f _bottom; float _0;
    aiVector3f _top;    float _1;
    union {
        struct { unsigned _idxLeft, _idxRight; unsigned int _2, _3; } inner;
        struct { unsigned _count, _startIndexList; unsigned int _type, _3; } leaf;   // _count: bit 31 = leaf flag
    } node;
};
struct DeviceTriangle { unsigned int idx0, idx1, idx2, primID; };
struct DeviceVertex   { float x, y, z, w; };

// ---- buffers bound to the raygen stage -------------------------------------------------------------
struct RayTraceProperties { unsigned int totalSamples, batchSize, depth, debug; };
struct Material {
    float albedo[4];
    float metallic, roughness, transmission, ior;
    int albedoTexIdx, metallicTexIdx, roughnessTexIdx, normalTexIdx;      // -1 = unused
};
struct MeshInfo { int vertexOffset, indexOffset, uvOffset, normalOffset, materialIndex, _0, _1, _2; };
struct DirLight { float direction[4]; float color[4]; };
struct SceneProperties { uint lightCount[4]; struct DirLight lights[5]; };   // only lightCount[0] is read
struct PhysicalCamera {
    float widthPixel, heightPixel;     // pixel counts as floats
    float focalLength, sensorWidth;    // metres
    float focalDistance, fStop;        // fStop 0 = pinhole
    float x, y, z;                     // position
    float wx, wy, wz;                  // rotation (radians, applied Rx * Ry * Rz)
};

static_assert(sizeof(DeviceBVHNode) == 48 && sizeof(DeviceInstance) == 80 && sizeof(Material) == 48 &&
              sizeof(SceneProperties) == 176 && sizeof(PhysicalCamera) == 48, "RD POD layout");
} // namespace RD

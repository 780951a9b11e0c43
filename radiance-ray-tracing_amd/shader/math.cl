/* math.cl -- matrix / random helpers of the device library (product-owned text).
 *
 * Same names, argument meaning and arithmetic as the reference's helper set (radiance/shader/math.cl:4-298), so a program
 * written against that library computes the same values: mat4x4 = float16, row-major (s0..s3 = row 0).
 */
#ifndef RDX_MATH_CL
#define RDX_MATH_CL

typedef float16 mat4x4;
typedef float3 vec3;
typedef float4 vec4;

/* PCG3D integer hash -> three floats in [0, 1] (math.cl:10-23); exact on every device */
float3 random_pcg3d(uint3 v)
{
    v = v * 1664525u + 1013904223u;
    v.x += v.y * v.z; v.y += v.z * v.x; v.z += v.x * v.y;
    v ^= v >> 16u;
    v.x += v.y * v.z; v.y += v.z * v.x; v.z += v.x * v.y;
    const float denom = (float)0xffffffffu;
    return (float3)((float)v.x / denom, (float)v.y / denom, (float)v.z / denom);
}

/* out = a * b, the four-term row sums in order (math.cl:25-31).  Like the reference's, the function reads its operands while it
 * writes `out` component by component: a caller that passes the same object for `b` and `out` gets the same (mixed) result. */
void MultiplyMat4Vec4(mat4x4* a, vec4* b, vec4* out)
{
    const float* A = (const float*)a;
    const float* B = (const float*)b;
    float* O = (float*)out;
    for (int r = 0; r < 4; ++r) O[r] = A[4 * r] * B[0] + A[4 * r + 1] * B[1] + A[4 * r + 2] * B[2] + A[4 * r + 3] * B[3];
}

/* out = a * b (math.cl:33-55): column by column, rows top to bottom, operands read live as above */
void MultiplyMat4Mat4(mat4x4* a, mat4x4* b, mat4x4* out)
{
    const float* A = (const float*)a;
    const float* B = (const float*)b;
    float* O = (float*)out;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            O[4 * r + c] = A[4 * r] * B[c] + A[4 * r + 1] * B[4 + c] + A[4 * r + 2] * B[8 + c] + A[4 * r + 3] * B[12 + c];
}

/* cofactor inverse, term order of math.cl:56-183; false (out untouched) when the determinant is 0 */
bool InverseMat4x4(mat4x4* mIn, mat4x4* invOut)
{
    float m[16], inv[16];
    vstore16(*mIn, 0, m);
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0) return false;
    det = 1.0f / det;
    for (int i = 0; i < 16; ++i) inv[i] = inv[i] * det;
    *invOut = vload16(0, inv);
    return true;
}

/* rotations about x / y / z by an angle in radians (math.cl:185-252) */
void EulerXToMat4x4(float t, mat4x4* out)
{
    const float c = cos(t), s = sin(t);
    *out = (mat4x4)(1.0f, 0.0f, 0.0f, 0.0f,   0.0f, c, -s, 0.0f,   0.0f, s, c, 0.0f,   0.0f, 0.0f, 0.0f, 1.0f);
}
void EulerYToMat4x4(float t, mat4x4* out)
{
    const float c = cos(t), s = sin(t);
    *out = (mat4x4)(c, 0.0f, s, 0.0f,   0.0f, 1.0f, 0.0f, 0.0f,   -s, 0.0f, c, 0.0f,   0.0f, 0.0f, 0.0f, 1.0f);
}
void EulerZToMat4x4(float t, mat4x4* out)
{
    const float c = cos(t), s = sin(t);
    *out = (mat4x4)(c, -s, 0.0f, 0.0f,   s, c, 0.0f, 0.0f,   0.0f, 0.0f, 1.0f, 0.0f,   0.0f, 0.0f, 0.0f, 1.0f);
}

void TransformToTranslate(mat4x4* a, vec3* out) { out->x = a->s3; out->y = a->s7; out->z = a->sb; }

void Vec4ToMat4x4(vec4 r0, vec4 r1, vec4 r2, vec4 r3, mat4x4* out) { *out = (mat4x4)(r0, r1, r2, r3); }

/* tangent frame [T B N] of a normal as the columns of a 4x4 (math.cl:269-298) */
void GetNormalSpace(float3 normal, mat4x4* out)
{
    const float3 xaxis = (float3)(1.0f, 0.0f, 0.0f);
    float3 tangent = (float3)(0.0f, 1.0f, 0.0f);
    if (1.0f - fabs(dot(xaxis, normal)) > 1e-6f) tangent = normalize(cross(xaxis, normal));
    const float3 bitangent = cross(normal, tangent);
    *out = (mat4x4)(tangent.x, bitangent.x, normal.x, 0.0f,
                    tangent.y, bitangent.y, normal.y, 0.0f,
                    tangent.z, bitangent.z, normal.z, 0.0f,
                    0.0f, 0.0f, 0.0f, 1.0f);
}

#endif

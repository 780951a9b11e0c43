/* user_shader.cl -- test program for the run-time shader path (csrc/user_shader.cpp): a self-contained OpenCL C `raygen`
 * with the reference's 14-parameter binding contract (samples/shader.cl:175-190) that proves every binding by writing values
 * derived from it.  Own code, no includes. */
struct Props { uint totalSamples, batchSize, depth, debug; };
struct Cam   { float widthPixel, heightPixel, focalLength, sensorWidth, focalDistance, fStop, x, y, z, wx, wy, wz; };

float wave(float a, float b) { return sin(a * 0.1f) * cos(b * 0.07f); }     /* goes through the OpenCL builtin library */

__kernel void raygen(__global struct Props* props, __global float* imageScratch, __global uchar* image,
                     __global struct Cam* cam, __global float* scene, __global int* meshInfo, __global float* vertexData,
                     __global uint* indexData, __global float* uvData, __global float* normalData, __global float* materials,
                     image2d_array_t textures, sampler_t sampler, __global uint* topLevel)
{
    const int i = get_global_id(0);
    const int W = (int)cam->widthPixel;
    const int x = i % W, y = i / W;
    imageScratch[4 * i + 0] = wave((float)x, (float)y);
    imageScratch[4 * i + 1] = (float)props->batchSize + vertexData[0] + normalData[1] + uvData[2];
    imageScratch[4 * i + 2] = (float)topLevel[0] + (float)indexData[1] + materials[0] + scene[4] + (float)meshInfo[4];
    image[4 * i + 0] = (uchar)(x & 255);
    image[4 * i + 1] = (uchar)(y & 255);
    image[4 * i + 2] = (uchar)(props->depth * 16u + topLevel[1]);
    image[4 * i + 3] = 255;
}

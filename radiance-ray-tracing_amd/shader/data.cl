/* data.cl -- device view of the acceleration-structure blob (product-owned text).
 *
 * Layout contract = the byte format RD::BuildAccelStruct produces and the reference's kernels read
 * (reference: radiance/shader/data.cl:4-99, radiance/src/core.h:34-101).  A top-level buffer is
 *     [ header 16 B | nodes 48 B each | instances 80 B each | bottom-level blobs ... ]
 * and a bottom-level blob is
 *     [ header 16 B | nodes 48 B each | triangles 16 B each | vertices float4 each ];
 * all offsets are bytes from the start of the blob they are stored in, except Instance.instanceOffset, which counts
 * from the start of the TOP-level buffer.  Only the names a user program can reach through the reference's library are
 * kept (struct AccelStruct as the opaque type of traceRay's first parameter, the type tags, Vertex); field names are
 * this file's own.
 */
#ifndef RDX_DATA_CL
#define RDX_DATA_CL

#define TYPE_INST 1
#define TYPE_TRIG 2
#define TYPE_TOP_AS 1
#define TYPE_BOT_AS 2

struct AccelStruct {            /* 16 B header of either kind */
    unsigned int type;          /* TYPE_TOP_AS / TYPE_BOT_AS */
    unsigned int nodeByteOffset;
    unsigned int secondOffset;  /* top: instances; bottom: triangles */
    unsigned int thirdOffset;   /* top: total buffer size; bottom: vertices */
};

struct BVHNode {                /* 48 B */
    float4 lo;                  /* box minimum, w unused */
    float4 hi;                  /* box maximum */
    unsigned int w0;            /* inner: left child;  leaf: 0x80000000 | count */
    unsigned int w1;            /* inner: right child; leaf: first instance / triangle */
    unsigned int w2;            /* leaf: TYPE_INST / TYPE_TRIG */
    unsigned int w3;
};

struct Triangle { unsigned int idx0, idx1, idx2, primID; };     /* 16 B */

struct Instance {               /* 80 B */
    float4 r0, r1, r2, r3;      /* rows of the object->world matrix */
    unsigned int SBTOffset;
    unsigned int instanceID;
    unsigned int customInstanceID;
    unsigned int instanceOffset;
};

struct RayTraceProperties { unsigned int totalSamples, batchSize, depth, debug; };

typedef float4 Vertex;

#define RDX_NODES(as)     ((__global const struct BVHNode*)(((__global const char*)(as)) + (as)->nodeByteOffset))
#define RDX_INSTANCES(as) ((__global const struct Instance*)(((__global const char*)(as)) + (as)->secondOffset))
#define RDX_TRIANGLES(as) ((__global const struct Triangle*)(((__global const char*)(as)) + (as)->secondOffset))
#define RDX_VERTICES(as)  ((__global const Vertex*)(((__global const char*)(as)) + (as)->thirdOffset))
#define RDX_IS_LEAF(n)    (((n)->w0 & 0x80000000u) != 0u)
#define RDX_COUNT(n)      ((n)->w0 & 0x7fffffffu)

#endif

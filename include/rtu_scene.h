/*
 * rtu_scene.h — flattened, pointer-free scene description handed across the
 * C-ABI boundary of the MI355X render path.
 *
 * This is the POD mirror of the reference's plugin surface (all citations are
 * relative to the reference tree):
 *   RtuNode      <- Node + Transformation      (ExternalLibrary/scene.h:223-262, 437-513)
 *   RtuMaterial  <- MtlBlinn                   (ExternalLibrary/materials.h:19-57)
 *   RtuLight     <- Ambient/Direct/PointLight  (ExternalLibrary/lights.h:28-100)
 *   RtuMesh      <- TriObj = cyTriMesh + cyBVH (ExternalLibrary/objects.h:45-66,
 *                                               cyTriMesh.h:106-123, cyBVH.h:187-203)
 *   RtuCamera    <- Camera                     (ExternalLibrary/scene.h:517-535)
 *   background / environment <- TexturedColor  (ExternalLibrary/scene.h:405-433)
 *
 * The node hierarchy is PRESERVED (pre-order array with parent links): the
 * reference re-transforms the ray level by level and re-normalises the normal
 * at every level (RenderFunctions.cpp:181-213), so composing transforms on the
 * host would change rounding and break bit parity.
 *
 * Plain C, no torch types, no pointers inside array elements.
 */
#ifndef RTU_SCENE_H_INCLUDED
#define RTU_SCENE_H_INCLUDED

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTU_BIGFLOAT 1.0e30f /* scene.h:55 */

/* Object type tag of a node (replaces the Object* virtual dispatch, scene.h:268-275). */
enum {
    RTU_OBJ_NONE    = 0, /* pure group node, e.g. Project4.xml "box" */
    RTU_OBJ_SPHERE  = 1, /* objects.h:21-28, unit sphere */
    RTU_OBJ_PLANE   = 2, /* objects.h:33-40, unit square z=0 */
    RTU_OBJ_TRIMESH = 3  /* objects.h:45-66 */
};

/* Light type tag (replaces Light virtual dispatch, scene.h:280-293). */
enum {
    RTU_LIGHT_AMBIENT = 0, /* lights.h:28-40 */
    RTU_LIGHT_DIRECT  = 1, /* lights.h:44-58 */
    RTU_LIGHT_POINT   = 2  /* lights.h:78-100 */
};

/* Limits of the device path. A scene beyond them is rejected by
 * rtu_upload_scene with RTU_ERR_UNSUPPORTED (never silently truncated). */
#define RTU_MAX_NODE_DEPTH   8   /* root = depth 0 */
#define RTU_MAX_BVH_STACK    48  /* reference: 100 (objFunctions.cpp:339); teapot needs 12 */
#define RTU_MAX_BOUNCE       5   /* RenderFunctions.cpp:134 passes bounceCount=5 */

/* One scene-graph node, pre-order. 128 bytes. */
typedef struct RtuNode {
    float   tm[9];        /* column-major 3x3, cyMatrix.h:290-294 */
    float   itm[9];       /* cached inverse, scene.h:228 */
    float   pos[3];       /* translation, scene.h:227 */
    int32_t parent;       /* index of parent node, -1 for the root */
    int32_t obj_type;     /* RTU_OBJ_* */
    int32_t mesh_id;      /* index into meshes[] when obj_type==RTU_OBJ_TRIMESH, else -1 */
    int32_t material_id;  /* index into materials[], -1 = node has no material */
    int32_t depth;        /* root = 0 */
    int32_t subtree_end;  /* one past the last descendant in pre-order */
    int32_t reserved[5];
} RtuNode;

/* MtlBlinn parameters: the plain colours of its TexturedColors; texture maps on them live in
 * RtuSceneDesc.material_maps (a map multiplies the colour, scene.h:421). 96 bytes. */
typedef struct RtuMaterial {
    float   diffuse[3];
    float   specular[3];
    float   reflection[3];
    float   refraction[3];
    float   emission[3];
    float   absorption[3];
    float   glossiness;
    float   ior;
    float   reflection_glossiness;
    float   refraction_glossiness;
    int32_t is_multi_fallback; /* 1: MultiMtl with no valid sub-material => Color(1,1,1), materials.h:66 */
    int32_t reserved;
} RtuMaterial;

/* 32 bytes. */
typedef struct RtuLight {
    int32_t type;          /* RTU_LIGHT_* */
    float   intensity[3];
    float   vec[3];        /* POINT: position; DIRECT: normalised direction */
    float   size;          /* POINT only; >0 => stochastic soft shadow */
} RtuLight;

/* BVH node, 32 bytes so a node is two aligned 16-byte loads and a sibling pair
 * is one 64-byte line. Same tree as cy::BVH (cyBVH.h:187-203) with the packed
 * word split into two fields:
 *   inner: count == 0, index = first child (second child = index+1)
 *   leaf : count in 1..8, index = offset into elements[] */
typedef struct RtuBvhNode {
    float    bmin[3];
    uint32_t index;
    float    bmax[3];
    uint32_t count;
} RtuBvhNode;

typedef struct RtuMesh {
    uint32_t nv, nf, nvn, nvt;       /* cyTriMesh.h:115-119 */
    uint32_t n_bvh_nodes;            /* including the unused node 0; root is node 1 (cyBVH.h:76) */
    uint32_t n_elements;             /* == nf */
    uint32_t bvh_depth;              /* number of levels, root = 1 */
    uint32_t reserved;
    float    bound_min[3];           /* cyTriMesh.h:122-123 */
    float    bound_max[3];
    const float*      v;             /* nv  * 3 */
    const uint32_t*   f;             /* nf  * 3 vertex indices */
    const float*      vn;            /* nvn * 3 */
    const uint32_t*   fn;            /* nf  * 3 normal indices */
    const float*      vt;            /* nvt * 3 (may be NULL when nvt==0) */
    const uint32_t*   ft;            /* nf  * 3 (may be NULL when nvt==0) */
    const RtuBvhNode* bvh;           /* n_bvh_nodes */
    const uint32_t*   elements;      /* n_elements face ids in leaf order */
} RtuMesh;

typedef struct RtuCamera {
    float   pos[3], dir[3], up[3];   /* after the loader fix-up, xmlload.cpp:108-126 */
    float   fov, focaldist, dof;
    int32_t img_width, img_height;
} RtuCamera;

/* Texture (scene.h:308-365): an image file sampled bilinearly with tiling (TextureFile,
 * texture.cpp:95-121) or a two-colour checker (TextureChecker, :125-133). */
enum { RTU_TEX_FILE = 0, RTU_TEX_CHECKER = 1 };
typedef struct RtuTexture {
    int32_t  type;                /* RTU_TEX_* */
    int32_t  width, height;       /* FILE: 0 x 0 when the file failed to load => samples black (texture.cpp:97) */
    int32_t  reserved;
    const uint8_t* rgb;           /* FILE: width*height Color24, rows as decoded (lodepng LCT_RGB / PPM P6) */
    float    color1[3], color2[3];/* CHECKER */
} RtuTexture;

/* TextureMap = Transformation + Texture* (scene.h:375-397). present == 0: the TexturedColor has no
 * map and samples its plain colour; present == 1 with texture == -1: TextureMap(NULL) => black. */
typedef struct RtuTexMap {
    int32_t present;
    int32_t texture;              /* index into textures[] or -1 */
    float   tm[9], itm[9], pos[3];/* TransformTo(p) = itm * (p - pos), scene.h:241 */
    int32_t reserved;
} RtuTexMap;                      /* 96 bytes */

/* The four TexturedColors of a MtlBlinn, in this order in RtuSceneDesc.material_maps[4*m + k]. */
enum { RTU_MAP_DIFFUSE = 0, RTU_MAP_SPECULAR = 1, RTU_MAP_REFLECTION = 2, RTU_MAP_REFRACTION = 3 };

/* TexturedColor reduced to what the in-scope configs need: a constant colour,
 * or "a texture map whose file failed to load" which samples black
 * (scene.h:382,421). */
typedef struct RtuEnvColor {
    float   color[3];
    int32_t has_map;       /* 1: a TextureMap is attached */
    int32_t map_is_null;   /* 1: TextureMap(NULL) => Sample() == black */
    int32_t reserved[3];
} RtuEnvColor;

typedef struct RtuSceneDesc {
    uint32_t n_nodes, n_materials, n_lights, n_meshes;
    const RtuNode*     nodes;       /* pre-order, nodes[0] is the root */
    const RtuMaterial* materials;
    const RtuLight*    lights;      /* in XML order == LightList order */
    const RtuMesh*     meshes;
    RtuCamera   camera;
    RtuEnvColor background;
    RtuEnvColor environment;
    /* textures ("next" row f2). n_textures == 0 and material_maps == NULL: an untextured scene. */
    uint32_t n_textures, reserved0;
    const RtuTexture* textures;
    const RtuTexMap*  material_maps;   /* NULL or n_materials * 4 */
    RtuTexMap background_map;          /* used when background.has_map && !map_is_null */
    RtuTexMap environment_map;
} RtuSceneDesc;

#ifdef __cplusplus
}
#endif
#endif /* RTU_SCENE_H_INCLUDED */

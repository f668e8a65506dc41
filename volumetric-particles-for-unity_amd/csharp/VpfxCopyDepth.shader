// VpfxCopyDepth.shader -- companion of MetavoxelManager.cs for OccluderSource.UnityDepthTextures (SOURCE ONLY: nothing here compiles ShaderLab).
// Copies a depth texture into an RFloat target so that it can be read back and handed to libvpfx:
//   pass 0: the light camera's depth map, raw D3D ortho depth (z - near) / (far - near): what FillVolume.shader:217-218 reads
//           (-> vp_fill_params.light_depth_map)
//   pass 1: the main camera's _CameraDepthTexture as LINEAR EYE DEPTH: what the ray-march's ZTest Less against mainSceneRT's depth
//           (VolumetricParticleRenderer.cs:204, RayMarchVoxel.shader:14) amounts to (-> vp_raymarch_params.scene_depth)
Shader "Hidden/Vpfx/CopyDepth" {
    Properties { _MainTex ("", 2D) = "white" {} }
    SubShader {
        ZTest Always Cull Off ZWrite Off
        Pass {
            CGPROGRAM
            #pragma vertex vert_img
            #pragma fragment frag
            #include "UnityCG.cginc"
            sampler2D_float _MainTex;
            float4 frag(v2f_img i) : SV_Target { return SAMPLE_DEPTH_TEXTURE(_MainTex, i.uv).xxxx; }
            ENDCG
        }
        Pass {
            CGPROGRAM
            #pragma vertex vert_img
            #pragma fragment frag
            #include "UnityCG.cginc"
            sampler2D_float _CameraDepthTexture;
            float4 frag(v2f_img i) : SV_Target
            {
                float raw = SAMPLE_DEPTH_TEXTURE(_CameraDepthTexture, i.uv);
                // nothing drawn: the library's "no geometry" value is any depth beyond the far plane
                return (raw >= 1.0 ? 3.0e38 : LinearEyeDepth(raw)).xxxx;
            }
            ENDCG
        }
    }
    Fallback Off
}

"""MetavoxelManager -- host-side mirror of the reference's `MetavoxelEngine.VolumetricParticleRenderer`
(Assets/Main Scene/VolumetricParticleRenderer.cs, "VPR.cs") for the hot path only.

Same inspector field names (VPR.cs:72-101), same entry points and call order:

    Start()                      VPR.cs:132   create resources (vp_create)
    OnPostRender(...)            VPR.cs:181   per-frame driver: gated bin+fill every `updateInterval` frames, ray-march
                                              every frame, composite over the scene
    UpdateMetavoxelPositions()   VPR.cs:370   vp_set_frame
    BinParticlesToMetavoxels()   VPR.cs:397   vp_bin
    FillMetavoxels()             VPR.cs:495   vp_fill
    RenderMetavoxels()           VPR.cs:637   vp_raymarch
    Set*() GUI setters           VPR.cs:1040-1119

This Python class exists because the container has no C# toolchain; `csharp/MetavoxelManager.cs` is the same
thing as the P/Invoke shim a Unity project would use.  All compute goes through libvpfx (no fallback).
"""
from __future__ import annotations

import math

import numpy as np

from . import abi
from .engine import Engine


class MetavoxelManager:
    def __init__(self, numMetavoxelsX=10, numMetavoxelsY=10, numMetavoxelsZ=10, mvScale=3.0, numVoxelsInMetavoxel=32,
                 numBorderVoxels=1, screenWidth=1024, screenHeight=768, device=-1):
        # ---- inspector fields (defaults = the demo scene, Volumetric_Particle_System.unity:9013-9026)
        self.numMetavoxelsX, self.numMetavoxelsY, self.numMetavoxelsZ = numMetavoxelsX, numMetavoxelsY, numMetavoxelsZ
        self.mvScale = float(mvScale)
        self.numVoxelsInMetavoxel = numVoxelsInMetavoxel
        self.numBorderVoxels = numBorderVoxels
        self.updateInterval = 2
        self.rayMarchSteps = 64
        self.ambientColor = (0.2, 0.2, 0.2)
        self.fDisplacementScale = 0.7
        self.fadeOutParticles = False
        self.opacityFactor = 0.04
        self.softParticleStepDistance = 20
        self.screenWidth, self.screenHeight, self.device = screenWidth, screenHeight, device
        # ---- scene bindings (dirLight, gridCenter, particleSys, displacement texture)
        self.lightToWorld = np.eye(4, dtype=np.float32).T.reshape(16).copy()
        self.wsGridCenter = np.zeros(3, dtype=np.float32)                       # VPR.cs:138
        self.psysLocalToWorld = np.eye(4, dtype=np.float32).T.reshape(16).copy()
        self.displacementCubemap = None                                         # float32 [6,S,S]
        self.lightDepthMap = None                                               # optional
        self.sceneDepth = None                                                  # optional
        self._engine = None
        self._frame_dirty = True
        self._cubemap_dirty = True
        self.numMetavoxelsCovered = 0
        self.particlesRT = None

    # ---- Unity callbacks ---------------------------------------------------------------------------------
    def Start(self):
        cfg = abi.vp_config()
        cfg.num_mv[0], cfg.num_mv[1], cfg.num_mv[2] = self.numMetavoxelsX, self.numMetavoxelsY, self.numMetavoxelsZ
        cfg.num_voxels, cfg.num_border, cfg.mv_scale = self.numVoxelsInMetavoxel, self.numBorderVoxels, self.mvScale
        cfg.width, cfg.height, cfg.device = self.screenWidth, self.screenHeight, self.device
        self._engine = Engine(cfg)
        self._frame_dirty = True

    def OnPostRender(self, frameCount, particles, layout, camera, mainSceneRT=None):
        """VPR.cs:181-220.  Returns particlesRT (and composites it over mainSceneRT in place when given)."""
        if frameCount % self.updateInterval == 0:                               # :186
            if self._frame_dirty:                                               # :188-195
                self.UpdateMetavoxelPositions()
            self.BinParticlesToMetavoxels(particles, layout)                    # :197
            self.FillMetavoxels()                                               # :198
        self.particlesRT = self.RenderMetavoxels(camera)                        # :207
        if mainSceneRT is not None:                                             # Blit(particlesRT, mainSceneRT, matBlendParticles) :210
            p = self.particlesRT
            mainSceneRT[..., :3] = p[..., :3] + mainSceneRT[..., :3] * (1.0 - p[..., 3:4])
            mainSceneRT[..., 3] = p[..., 3] + mainSceneRT[..., 3]
        return self.particlesRT

    # ---- the hot path ------------------------------------------------------------------------------------
    def UpdateMetavoxelPositions(self):
        self._engine.set_frame(self.lightToWorld, self.wsGridCenter)
        self._frame_dirty = False

    def BinParticlesToMetavoxels(self, particles, layout):
        if self._frame_dirty:
            self.UpdateMetavoxelPositions()
        self._engine.bin(particles, layout, self.psysLocalToWorld)

    def FillMetavoxels(self):
        if self.displacementCubemap is None:
            raise ValueError("displacementCubemap is not set (FillVolume.mat binds _DisplacementTexture)")
        p = abi.vp_fill_params()
        p.opacity_factor = self.opacityFactor
        p.displacement_scale = self.fDisplacementScale
        p.fade_out_particles = 1 if self.fadeOutParticles else 0
        p.ambient[0], p.ambient[1], p.ambient[2] = self.ambientColor
        p.init_light_intensity = 1.0                                            # VPR.cs:540
        p.light_near, p.light_far, p.light_cam_distance = 0.3, 1000.0, 200.0    # VPR.cs:342,365
        cube = np.ascontiguousarray(self.displacementCubemap, dtype=np.float32)
        p.cubemap_size = cube.shape[1]
        if self._cubemap_dirty:
            p.cubemap = cube.ctypes.data_as(abi.c_float_p)                      # uploaded once, then resident
        depth = None
        if self.lightDepthMap is not None:
            depth = np.ascontiguousarray(self.lightDepthMap, dtype=np.float32)
            p.light_depth_map = depth.ctypes.data_as(abi.c_float_p)
        self._engine.fill(p)
        self._cubemap_dirty = False
        self.numMetavoxelsCovered = self._engine.stats()["occupied_mv"]         # VPR.cs:515

    def RenderMetavoxels(self, camera):
        rp = abi.vp_raymarch_params()
        rp.steps_per_mv = self.rayMarchSteps
        rp.soft_distance = self.softParticleStepDistance
        depth = None
        if self.sceneDepth is not None:
            depth = np.ascontiguousarray(self.sceneDepth, dtype=np.float32)
            rp.scene_depth = depth.ctypes.data_as(abi.c_float_p)
        return self._engine.raymarch(camera, rp)

    # ---- scene bindings ----------------------------------------------------------------------------------
    def SetLight(self, lightToWorld16):
        self.lightToWorld = np.ascontiguousarray(lightToWorld16, dtype=np.float32)
        self._frame_dirty = True                                                # light rotated: VPR.cs:188

    def SetGridCenter(self, pos):
        self.wsGridCenter = np.ascontiguousarray(pos, dtype=np.float32)
        self._frame_dirty = True                                                # gridCenter moved: VPR.cs:189

    def SetOccluders(self, boxes):
        """Opaque scene geometry as boxes: the light depth map and the eye depth are then rendered on the GPU
        (what lightCamera.RenderWithShader and the main camera's depth buffer provide in the reference, VPR.cs:184,204)."""
        self._engine.set_occluders(list(boxes))

    def SetDisplacementTexture(self, cubemap):
        self.displacementCubemap = cubemap
        self._cubemap_dirty = True

    # ---- GUI setters (VPR.cs:1040-1119) ------------------------------------------------------------------
    def SetOpacityFactor(self, v): self.opacityFactor = float(v)
    def SetDisplacementScale(self, v): self.fDisplacementScale = float(v)
    def SetRayMarchSteps(self, v): self.rayMarchSteps = int(v)
    def SetSoftParticleStepDistance(self, v): self.softParticleStepDistance = int(v)
    def SetUpdateInterval(self, v): self.updateInterval = max(1, int(v))
    def SetFadeOutParticles(self, v): self.fadeOutParticles = bool(v)
    def SetAmbientColor(self, rgb): self.ambientColor = tuple(float(x) for x in rgb)

    @staticmethod
    def MakeCamera(worldToCamera16, cameraToWorld16, position, fieldOfViewDeg, near=0.3, far=1000.0):
        cam = abi.vp_camera()
        for i in range(16):
            cam.world_to_camera[i] = worldToCamera16[i]
            cam.camera_to_world[i] = cameraToWorld16[i]
        for i in range(3):
            cam.cam_pos[i] = position[i]
        cam.fov_y = math.radians(fieldOfViewDeg)                                # Mathf.Deg2Rad * fieldOfView  VPR.cs:734
        cam.near_clip, cam.far_clip = near, far
        return cam

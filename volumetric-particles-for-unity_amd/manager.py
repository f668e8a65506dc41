"""MetavoxelManager -- host-side mirror of the reference's `MetavoxelEngine.VolumetricParticleRenderer`
(Assets/Main Scene/VolumetricParticleRenderer.cs, "VPR.cs") for the hot path only.

Same inspector field names (VPR.cs:72-101), same entry points and call order:

    Start()                      VPR.cs:132   create resources (vp_create)
    OnPreRender(mainSceneRT)     VPR.cs:168   clear the scene target (particlesRT lives in the library: vp_raymarch starts from 0)
    OnPostRender(...)            VPR.cs:181   per-frame driver: the opaque scene for the two depth inputs (SyncOccluders), gated bin+fill
                                              every `updateInterval` frames, ray-march every frame, composite over the scene
    SyncOccluders()              VPR.cs:184   vp_set_occluders2 when the solids changed (the library renders the light depth map inside
                                              vp_fill and the eye depth inside vp_raymarch); or lightDepthMap / sceneDepth arrays passed as pointers
    CompositeParticles(rt)       VPR.cs:210   CompositeParticles.shader:10
    UpdateMetavoxelPositions()   VPR.cs:370   vp_set_frame
    BinParticlesToMetavoxels()   VPR.cs:397   vp_bin
    FillMetavoxels()             VPR.cs:495   vp_fill               (one persistent launch)
    FillMetavoxel(xx, yy, zz)    VPR.cs:559   vp_fill_metavoxel     (the reference's per-metavoxel draw, for debugging / replays)
    RenderMetavoxels()           VPR.cs:637   vp_raymarch           (one launch)
    RenderMetavoxel(xx,yy,zz,i)  VPR.cs:766   vp_render_metavoxel   (one metavoxel blended into particlesRT)
    FillMetavoxelsPerDraw() / RenderMetavoxelsPerDraw(): the reference's literal loops over those two (VPR.cs:505-518, 652-711)
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
                 numBorderVoxels=1, screenWidth=1024, screenHeight=768, device=-1, gpuDevices=(), multiFlags=0, rebalanceInterval=240):
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
        # ---- added by this binding (same names as csharp/MetavoxelManager.cs): the device list of a multi-GPU fan-out context
        self.gpuDevices, self.multiFlags, self.rebalanceInterval = tuple(gpuDevices), multiFlags, rebalanceInterval
        self.asyncReadback = False          # vp_raymarch_async: RenderMetavoxels returns LAST frame's image (None on the first frame; a view of one of
                                            # two read-back buffers, valid until the call after next), the copy of this frame's image runs beside the
                                            # next frame's bin + fill
        self._async_bufs, self._async_i, self._async_pending = None, 0, False
        # ---- scene bindings (dirLight, gridCenter, particleSys, displacement texture)
        self.lightToWorld = np.eye(4, dtype=np.float32).T.reshape(16).copy()
        self.wsGridCenter = np.zeros(3, dtype=np.float32)                       # VPR.cs:138
        self.psysLocalToWorld = np.eye(4, dtype=np.float32).T.reshape(16).copy()
        self.displacementCubemap = None                                         # float32 [6,S,S]
        self.lightDepthMap = None                                               # optional: the C# shim's OccluderSource.UnityDepthTextures
        self.sceneDepth = None                                                  # optional:   "
        self._occluders, self._occluders_dirty = [], False                      # the C# shim's OccluderSource.SceneMeshes
        self._engine = None
        self._frame_dirty = True
        self._cubemap_dirty = True
        self.numMetavoxelsCovered = 0
        self.particlesRT = None
        self._filled_once = False
        self.showMetavoxelDrawOrder = False                                     # VPR.cs:754

    # ---- Unity callbacks ---------------------------------------------------------------------------------
    def Start(self):
        cfg = abi.vp_config()
        cfg.num_mv[0], cfg.num_mv[1], cfg.num_mv[2] = self.numMetavoxelsX, self.numMetavoxelsY, self.numMetavoxelsZ
        cfg.num_voxels, cfg.num_border, cfg.mv_scale = self.numVoxelsInMetavoxel, self.numBorderVoxels, self.mvScale
        cfg.width, cfg.height, cfg.device = self.screenWidth, self.screenHeight, self.device
        if len(self.gpuDevices) > 1:                                            # the ONE thing a multi-GPU host adds
            cfg.num_devices = len(self.gpuDevices)
            for i, d in enumerate(self.gpuDevices):
                cfg.devices[i] = d
            cfg.multi_flags = self.multiFlags
        elif len(self.gpuDevices) == 1:
            cfg.device = self.gpuDevices[0]
        self._engine = Engine(cfg)
        self._frame_dirty = True

    def OnDestroy(self):
        """Release the read-back buffers' page locks and the native context (the C# shim's OnDestroy)."""
        eng = self._engine
        if eng is not None and getattr(eng, "h", None):
            if self._async_bufs is not None:
                eng.sync()                                  # lands a pending image before its buffer loses the page lock
                for b in self._async_bufs:
                    eng.unpin(b)
            eng.close()
        self._async_bufs, self._async_pending, self._engine = None, False, None

    def __del__(self):
        try:
            self.OnDestroy()
        except Exception:
            pass

    def OnPreRender(self, mainSceneRT=None):
        """VPR.cs:168-177.  particlesRT lives in the library and vp_raymarch starts from dst = 0 (the per-metavoxel path clears it with
        vp_clear_particles_rt, see RenderMetavoxelsPerDraw); mainSceneRT is the host's: cleared to Color.black."""
        if mainSceneRT is not None:
            mainSceneRT[..., :3] = 0.0
            mainSceneRT[..., 3] = 1.0

    def OnPostRender(self, frameCount, particles, layout, camera, mainSceneRT=None):
        """VPR.cs:181-220.  Returns particlesRT (and composites it over mainSceneRT in place when given)."""
        # the reference's first OnPostRender has frameCount % updateInterval == 0 (frame 0); a host that starts on another frame
        # would ray-march textures that were never filled, so the first call always bins + fills
        if len(self.gpuDevices) > 1 and self.rebalanceInterval > 0 and frameCount % self.rebalanceInterval == 0 and self._filled_once:
            self._engine.rebalance()                                            # re-cut the slabs from the work the GPUs measured
        self.SyncOccluders()                                                    # :184 the light's (and the eye's) view of the opaque scene
        if frameCount % self.updateInterval == 0 or not self._filled_once:      # :186
            if self._frame_dirty:                                               # :188-195
                self.UpdateMetavoxelPositions()
            self.BinParticlesToMetavoxels(particles, layout)                    # :197
            self.FillMetavoxels()                                               # :198
        self.particlesRT = self.RenderMetavoxels(camera)                        # :204-207
        if self.particlesRT is None:                                            # asyncReadback, first frame: nothing to show yet
            return None
        self.CompositeParticles(mainSceneRT)                                    # :210
        return self.particlesRT

    def CompositeParticles(self, mainSceneRT):
        """Blit(particlesRT, mainSceneRT, matBlendParticles) (VPR.cs:210; Comp.shader:10: Blend One OneMinusSrcAlpha, One One)."""
        if mainSceneRT is None or self.particlesRT is None:
            return
        p = self.particlesRT
        mainSceneRT[..., :3] = p[..., :3] + mainSceneRT[..., :3] * (1.0 - p[..., 3:4])
        mainSceneRT[..., 3] = p[..., 3] + mainSceneRT[..., 3]

    def SyncOccluders(self):
        """Hand the opaque scene's solids to the library when they changed (the C# shim walks the Default-layer MeshRenderers here)."""
        if self._occluders_dirty:
            self._engine.set_occluders2(self._occluders)
            self._occluders_dirty = False

    # ---- the hot path ------------------------------------------------------------------------------------
    def UpdateMetavoxelPositions(self):
        self._engine.set_frame(self.lightToWorld, self.wsGridCenter)
        self._frame_dirty = False

    def BinParticlesToMetavoxels(self, particles, layout):
        if self._frame_dirty:
            self.UpdateMetavoxelPositions()
        self._engine.bin(particles, layout, self.psysLocalToWorld)

    def _fill_params(self):
        if self.displacementCubemap is None:
            raise ValueError("displacementCubemap is not set (FillVolume.mat binds _DisplacementTexture)")
        p = abi.vp_fill_params()
        p.opacity_factor = self.opacityFactor
        p.displacement_scale = self.fDisplacementScale
        p.fade_out_particles = 1 if self.fadeOutParticles else 0
        p.ambient[0], p.ambient[1], p.ambient[2] = self.ambientColor
        p.init_light_intensity = 1.0                                            # VPR.cs:540
        p.light_near, p.light_far, p.light_cam_distance = 0.3, 1000.0, 200.0    # VPR.cs:342,365
        # float32 [6,S,S] in [0,1], or uint8 (R8: the reference's own asset is an 8-bit texture, DisplacementTexture.cubemap:10-23)
        cube = np.ascontiguousarray(self.displacementCubemap)
        if cube.dtype != np.uint8:
            cube = np.ascontiguousarray(cube, dtype=np.float32)
        abi.set_cubemap(p, cube)
        if not self._cubemap_dirty:
            p.cubemap = None                                                    # uploaded once, then resident
        depth = None
        if self.lightDepthMap is not None:
            depth = np.ascontiguousarray(self.lightDepthMap, dtype=np.float32)
            p.light_depth_map = depth.ctypes.data_as(abi.c_float_p)
        return p, (cube, depth)                                                 # keep the arrays alive for the call

    def FillMetavoxels(self):
        p, keep = self._fill_params()
        self._engine.fill(p)
        self._cubemap_dirty = False                                             # only after a fill that succeeded (it raises otherwise)
        self._filled_once = True
        self.numMetavoxelsCovered = self._engine.stats()["occupied_mv"]         # VPR.cs:515

    def FillMetavoxel(self, xx, yy, zz):
        """VPR.cs:559-609: fill ONE metavoxel (light in/out through the light-propagation map).  Needs FillMetavoxelsBegin()."""
        self._engine.fill_metavoxel(xx, yy, zz)

    def FillMetavoxelsBegin(self):
        """Head of FillMetavoxels (VPR.cs:497-503): SetFillPassConstants + clear of lightPropogationTex."""
        p, keep = self._fill_params()
        self._engine.fill_begin(p)
        self._cubemap_dirty = False

    def FillMetavoxelsPerDraw(self):
        """The reference's literal loop: zz-major, then yy, then xx, one FillMetavoxel per occupied metavoxel (VPR.cs:505-518)."""
        self.FillMetavoxelsBegin()
        counts = self._engine.bin_counts()
        self.numMetavoxelsCovered = 0
        for zz in range(self.numMetavoxelsZ):
            for yy in range(self.numMetavoxelsY):
                for xx in range(self.numMetavoxelsX):
                    if counts[zz, yy, xx] != 0:                                 # mParticlesCovered.Count != 0   :511
                        self.FillMetavoxel(xx, yy, zz)
                        self.numMetavoxelsCovered += 1
        self._filled_once = True

    def _raymarch_params(self):
        rp = abi.vp_raymarch_params()
        rp.steps_per_mv = self.rayMarchSteps
        rp.soft_distance = self.softParticleStepDistance
        if self.showMetavoxelDrawOrder:
            rp.flags |= abi.VP_RM_SHOW_DRAW_ORDER
        depth = None
        if self.sceneDepth is not None:
            depth = np.ascontiguousarray(self.sceneDepth, dtype=np.float32)
            rp.scene_depth = depth.ctypes.data_as(abi.c_float_p)
        return rp, depth

    def RenderMetavoxels(self, camera):
        rp, keep = self._raymarch_params()
        if not self.asyncReadback:
            return self._engine.raymarch(camera, rp)
        if self._async_bufs is None:
            self._async_bufs = [np.zeros((self._engine.H, self._engine.W, 4), dtype=np.float32) for _ in range(2)]
            for b in self._async_bufs:
                self._engine.pin(b)
        last = None
        if self._async_pending:
            self._engine.wait_image()                                           # last frame's image has landed
            last = self._async_bufs[self._async_i]
        self._async_i ^= 1
        self._engine.raymarch_async(camera, rp, self._async_bufs[self._async_i])
        self._async_pending = True
        return last

    def RenderMetavoxel(self, camera, xx, yy, zz, orderIndex=0, blendOver=False):
        """VPR.cs:766-794: one metavoxel marched and blended into particlesRT with the blend state of its phase."""
        rp, keep = self._raymarch_params()
        self._engine.render_metavoxel(camera, rp, xx, yy, zz, blendOver, orderIndex)

    def RenderMetavoxelsPerDraw(self, camera):
        """The reference's literal submission loop (VPR.cs:637-713) over RenderMetavoxel; returns particlesRT."""
        e = self._engine
        counts = e.bin_counts()
        pos = e.mv_positions()
        cam_pos = np.array([camera.cam_pos[i] for i in range(3)], dtype=np.float32)
        # SortMetavoxelSlicesFarToNearFromEye: keys from slice zz = 0, list built yy-major, stable ascending sort, reversed (VPR.cs:613-632)
        d = pos[0].reshape(-1, 3) - cam_pos
        key = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        near_to_far = np.argsort(key, kind="stable")
        zb = e.z_boundary(camera)                                               # VPR.cs:642-648
        e.clear_particles_rt()                                                  # OnPreRender                 VPR.cs:171
        mvCount = 0
        for zz in range(0, zb + 1):                                             # phase A: far -> near, OVER  VPR.cs:667-681
            for c in near_to_far[::-1]:
                xx, yy = int(c % self.numMetavoxelsX), int(c // self.numMetavoxelsX)
                if counts[zz, yy, xx] != 0:
                    self.RenderMetavoxel(camera, xx, yy, zz, mvCount, blendOver=True)
                    mvCount += 1
        for zz in range(zb + 1, self.numMetavoxelsZ):                           # phase B: near -> far, UNDER VPR.cs:695-711
            for c in near_to_far:
                xx, yy = int(c % self.numMetavoxelsX), int(c // self.numMetavoxelsX)
                if counts[zz, yy, xx] != 0:
                    self.RenderMetavoxel(camera, xx, yy, zz, mvCount, blendOver=False)
                    mvCount += 1
        return e.read_particles_rt()

    # ---- scene bindings ----------------------------------------------------------------------------------
    def SetLight(self, lightToWorld16):
        self.lightToWorld = np.ascontiguousarray(lightToWorld16, dtype=np.float32)
        self._frame_dirty = True                                                # light rotated: VPR.cs:188

    def SetGridCenter(self, pos):
        self.wsGridCenter = np.ascontiguousarray(pos, dtype=np.float32)
        self._frame_dirty = True                                                # gridCenter moved: VPR.cs:189

    def SetOccluders(self, solids):
        """Opaque scene geometry as solids (abi.vp_obb boxes / abi.vp_occluder boxes, cylinders, ellipsoids): the light depth map and the eye
        depth are then rendered on the GPU (what lightCamera.RenderWithShader and the main camera's depth buffer provide in the reference,
        VPR.cs:184,204).  Sent by the next SyncOccluders (= the next OnPostRender)."""
        self._occluders, self._occluders_dirty = list(solids), True

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
    def SetShowMetavoxelDrawOrder(self, v): self.showMetavoxelDrawOrder = bool(v)   # VPR.cs:1096-1099 -> _ShowMetavoxelDrawOrder (:750-754)
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

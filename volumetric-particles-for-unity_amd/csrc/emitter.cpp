// emitter.cpp -- the particle source of the reference's demo scene behind the C ABI (include/vpfx.h "particle source").
//
// The reference feeds the hot path from a Unity ParticleSystem ("Particle System Demo", Assets/Volumetric_Particle_System.unity:2264-2620):
// InitialModule :2272-2497 (lifetime 6 s, speed 3, size 4, random start rotation, at most 60 particles: the inspector slider of VPR.cs),
// ShapeModule :2498-2511 (type 4 = cone, angle 10 deg, radius 0.5), EmissionModule :2512-2556 (10 particles / s), RotationModule :2587-2621
// (angular velocity 0.0698 rad/s = 4 deg/s); simulated in the system's LOCAL space, which is what ParticleSystem.GetParticles hands to
// BinParticlesToMetavoxels (VPR.cs:412-420).  Unity's emitter and its random stream are closed source: this reproduces the documented
// parameters with an own, fully specified generator (PCG32, XSH-RR 64/32) and f32 arithmetic, so that a headless host (examples/, bench.py
// --config DEMO, the tests) gets temporally coherent input in the ParticleSystem.Particle layout the C ABI consumes -- the counterpart of
// a Unity scene's particle system, not of anything on the GPU.  Host code only: a handful of particles per frame.
#include "../../include/vpfx.h"

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

struct vp_emitter {
    vp_emitter_config cfg;
    uint64_t state, inc;              // PCG32
    float acc;                        // fractional particles owed by the emission rate
    double time;
    struct P { float pos[3], vel[3], rot_deg, life; };
    std::vector<P> live;              // in order of birth (the oldest first)
};

namespace {

inline uint32_t pcg32(vp_emitter* e)
{
    const uint64_t old = e->state;
    e->state = old * 6364136223846793005ULL + e->inc;
    const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    const uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}
// uniform in [0, 1): the top 24 bits, exact in f32
inline float uniform01(vp_emitter* e) { return (float)(pcg32(e) >> 8) * (1.0f / 16777216.0f); }

inline void put_f32(unsigned char* rec, int32_t off, float v) { std::memcpy(rec + off, &v, 4); }

bool layout_ok(const vp_particle_layout* l)
{
    if (!l || l->stride < 4) return false;
    const int32_t offs[5] = {l->off_position, l->off_size, l->off_rotation, l->off_lifetime, l->off_start_lifetime};
    for (int i = 0; i < 5; ++i) {
        const int32_t bytes = i == 0 ? 12 : 4;
        if (offs[i] < 0 || offs[i] + bytes > l->stride) return false;
    }
    return true;
}

}  // namespace

extern "C" {

#define VP_EMIT_API __attribute__((visibility("default")))

VP_EMIT_API void vp_emitter_default_config(vp_emitter_config* cfg)
{
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->seed = 7;
    cfg->rate = 10.0f;                 // EmissionModule, scene:2512-2556
    cfg->lifetime = 6.0f;              // InitialModule, scene:2272-2497
    cfg->speed = 3.0f;
    cfg->size = 4.0f;
    cfg->cone_angle_deg = 10.0f;       // ShapeModule type 4, scene:2498-2511
    cfg->cone_radius = 0.5f;
    cfg->angular_velocity_deg = 4.0f;  // RotationModule 0.0698 rad/s, scene:2587-2621
    cfg->max_particles = 60;           // numParticlesEmitted slider, VPR.cs (0..128), scene default 60
}

VP_EMIT_API int vp_emitter_create(const vp_emitter_config* cfg, vp_emitter** out)
{
    if (!cfg || !out) return VP_ERR_BAD_ARG;
    *out = nullptr;
    // lifetime: finite and <= 1e5 s (the prewarm below simulates lifetime x 30 steps: +inf / 1e30 would be an endless loop or a UB cast)
    if (!(cfg->rate >= 0.f) || !std::isfinite(cfg->rate) || !(cfg->lifetime > 0.f) || !(cfg->lifetime <= 1.0e5f) || !(cfg->cone_radius > 0.f) || !(cfg->cone_angle_deg >= 0.f && cfg->cone_angle_deg < 90.f) ||
        !(cfg->size > 0.f) || !std::isfinite(cfg->speed) || !std::isfinite(cfg->angular_velocity_deg) || cfg->max_particles < 0 ||
        cfg->max_particles > (1 << 24))
        return VP_ERR_BAD_ARG;
    vp_emitter* e = new (std::nothrow) vp_emitter();
    if (!e) return VP_ERR_OOM;
    e->cfg = *cfg;
    e->acc = 0.f;
    e->time = 0.0;
    // PCG32 seeding (stream constant fixed): state = 0, step, add the seed, step
    e->state = 0;
    e->inc = (0xda3e39cb94b95bdbULL << 1u) | 1u;
    pcg32(e);
    e->state += cfg->seed;
    pcg32(e);
    // prewarm (the reference's system has `prewarm: 1`, scene:2269: it starts in steady state, ~rate x lifetime particles alive): simulate one
    // lifetime in 1/30 s steps before the first frame
    if (cfg->reserved[0] == 1) {
        const int steps = (int)std::ceil((double)cfg->lifetime * 30.0);
        for (int i = 0; i < steps; ++i) { const int rc = vp_emitter_step(e, 1.0f / 30.0f); if (rc < 0) { delete e; return rc; } }
    } else if (cfg->reserved[0] != 0) { delete e; return VP_ERR_BAD_ARG; }
    *out = e;
    return VP_OK;
}

VP_EMIT_API void vp_emitter_destroy(vp_emitter* e) { delete e; }

// One simulation step of dt seconds: move, age and retire the live particles, then emit what the rate owes (capped by max_particles).
// Returns the live count (>= 0) or a negative vp_status.
VP_EMIT_API int vp_emitter_step(vp_emitter* e, float dt)
{
    if (!e || !(dt >= 0.f) || !std::isfinite(dt)) return VP_ERR_BAD_ARG;
    const vp_emitter_config& c = e->cfg;
    e->time += dt;
    size_t w = 0;
    for (size_t i = 0; i < e->live.size(); ++i) {
        vp_emitter::P p = e->live[i];
        p.pos[0] += p.vel[0] * dt; p.pos[1] += p.vel[1] * dt; p.pos[2] += p.vel[2] * dt;
        p.rot_deg += c.angular_velocity_deg * dt;
        p.life -= dt;
        if (p.life > 0.f) e->live[w++] = p;
    }
    e->live.resize(w);
    e->acc += c.rate * dt;
    if (!(e->acc < 16777216.0f)) e->acc = 16777216.0f;      // (more than max_particles can ever take; keeps the conversion below defined)
    const int owed = (int)e->acc;
    e->acc -= (float)owed;
    const int room = c.max_particles - (int)e->live.size();
    const int n = owed < room ? owed : room;
    const float cone = c.cone_angle_deg * 0.017453292519943295f;
    for (int i = 0; i < n; ++i) {
        // a point of the cone's base disc (uniform by area); a point at the rim leaves along the cone's surface, the centre along the axis (+z)
        const float u = uniform01(e), v = uniform01(e), wrot = uniform01(e);
        const float rn = std::sqrt(u);                                 // r / radius
        const float phi = 6.283185307179586f * v;
        const float cp = std::cos(phi), sp = std::sin(phi);
        const float tilt = cone * rn;
        const float st = std::sin(tilt), ct = std::cos(tilt);
        vp_emitter::P p;
        p.pos[0] = c.cone_radius * rn * cp; p.pos[1] = c.cone_radius * rn * sp; p.pos[2] = 0.f;
        p.vel[0] = c.speed * (st * cp); p.vel[1] = c.speed * (st * sp); p.vel[2] = c.speed * ct;
        p.rot_deg = 360.0f * wrot;
        p.life = c.lifetime;
        try { e->live.push_back(p); } catch (...) { return VP_ERR_OOM; }
    }
    return (int)e->live.size();
}

VP_EMIT_API int vp_emitter_count(const vp_emitter* e) { return e ? (int)e->live.size() : VP_ERR_BAD_ARG; }

// The live particles as ParticleSystem.Particle records of the caller's layout (what GetParticles fills, VPR.cs:412-413): every record is
// zeroed, then position (local space), size, rotation (degrees, or radians when layout->rotation_in_radians), lifetime (remaining) and
// startLifetime are written at the layout's offsets -- the five fields the path reads (VPR.cs:418-425,583-586).  Returns the number of records
// written (<= capacity; the oldest particles first) or a negative vp_status.
VP_EMIT_API int vp_emitter_write_particles(const vp_emitter* e, void* particles_out, int32_t capacity, const vp_particle_layout* layout)
{
    if (!e || capacity < 0 || (capacity > 0 && !particles_out) || !layout_ok(layout)) return VP_ERR_BAD_ARG;
    const int n = (int)e->live.size() < capacity ? (int)e->live.size() : capacity;
    unsigned char* rec = static_cast<unsigned char*>(particles_out);
    for (int i = 0; i < n; ++i, rec += layout->stride) {
        const vp_emitter::P& p = e->live[i];
        std::memset(rec, 0, (size_t)layout->stride);
        put_f32(rec, layout->off_position, p.pos[0]); put_f32(rec, layout->off_position + 4, p.pos[1]); put_f32(rec, layout->off_position + 8, p.pos[2]);
        put_f32(rec, layout->off_size, e->cfg.size);
        const float deg = std::fmod(p.rot_deg, 360.0f);
        put_f32(rec, layout->off_rotation, layout->rotation_in_radians ? deg * 0.017453292519943295f : deg);
        put_f32(rec, layout->off_lifetime, p.life);
        put_f32(rec, layout->off_start_lifetime, e->cfg.lifetime);
    }
    return n;
}

}  // extern "C"

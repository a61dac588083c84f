// illuminant_host.hpp -- host-side mirror of the reference's C# interface for the two hot paths.
//
// The reference's host code is C# (.NET 4.8); no .NET toolchain exists in the build image, so the host
// side above the C ABI is written in C++ with the reference's own class / member names, argument meaning
// and error behaviour (exceptions with the same messages).  It uses ONLY include/illuminant_hip.h --
// exactly what the C# P/Invoke layer of INTEGRATION.md would call -- and contains no device code.
//
// Mirrors (file:line relative to the reference checkout):
//   Particles::ParticleEngine / ParticleEngineConfiguration   Illuminant/Particles/ParticleEngine.cs:24-141,616-696
//   Particles::ParticleSystem (+Chunk, liveness, spawning)     Illuminant/Particles/ParticleSystem.cs:48-1072,
//                                                              ParticleSpawning.cs:13-231, ParticleLiveness.cs:14-129
//   Particles::Transforms::{Spawner,Gravity,Noise,FMA}         Illuminant/Particles/ParticleSpawner.cs:16-419, Transforms.cs:16-372
//   DistanceField                                              Illuminant/SDF/DistanceField.cs:18-246
//   Lighting::{LightingEnvironment,SphereLightSource,LightingRenderer,RendererQualitySettings}
//                                                              Illuminant/Lighting/LightingEnvironment.cs:13-49, LightSource.cs:37-280,
//                                                              LightingRenderer.cs:917-1219,1894-1940, LightingRenderer.Configuration.cs:254-291
// Parameter<T> animation (bezier / named / expression parameters, Parameter.cs:190-485) is evaluated on the
// CPU before upload in the reference; here every parameter is a plain value (out of scope).
#pragma once

#include <algorithm>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/illuminant_hip.h"

namespace Squared {
namespace Illuminant {

struct Vector2 { float X = 0, Y = 0; };
struct Vector3 { float X = 0, Y = 0, Z = 0; };
struct Vector4 { float X = 0, Y = 0, Z = 0, W = 0; };

// .NET exception types the reference throws on these paths
struct InvalidOperationException : std::runtime_error { using std::runtime_error::runtime_error; };
struct ArgumentException : std::invalid_argument { using std::invalid_argument::invalid_argument; };
// anything the native layer reports (hipError_t or ILM_ERR_*)
struct NativeException : std::runtime_error {
    int Code;
    NativeException(int code, const std::string& what) : std::runtime_error(what), Code(code) {}
};
void ThrowIfFailed(int32_t code);

// Stand-in for Squared.CoreCLR.Xoshiro (Fracture, not in the reference tree): xoshiro256** seeded through
// SplitMix64.  Every draw the reference makes from its RNG is an explicit value at the C ABI, so the
// generator only has to be deterministic, not identical.
class Xoshiro {
public:
    explicit Xoshiro(uint64_t seed = 0x1234567ull);
    uint64_t NextUInt64();
    double NextDouble();   // [0, 1)
private:
    uint64_t s[4];
};

struct ITimeProvider {
    virtual ~ITimeProvider() {}
    virtual double Seconds() = 0;
};
class ManualTimeProvider : public ITimeProvider {
public:
    double Now = 0;
    double Seconds() override { return Now; }
    void Advance(double dt) { Now += dt; }
};

// One GPU context (replaces the RenderCoordinator / GraphicsDevice the reference threads through)
class DeviceContext {
public:
    explicit DeviceContext(int deviceId = 0);
    // a context owned by someone else -- a member of a multi-device group (ilm_group_ctx): used, never destroyed, by this object
    struct Borrowed { IlmHandle handle; };
    explicit DeviceContext(Borrowed member) : handle(member.handle), owned(false) {}
    ~DeviceContext();
    DeviceContext(const DeviceContext&) = delete;
    IlmHandle Handle() const { return handle; }
    void Sync();
    void TimerStart();
    float TimerStop();
private:
    IlmHandle handle = 0;
    bool owned = true;
};

// ---- SDF/DistanceField.cs ---------------------------------------------------------------------------
// SliceInfo, :13-16
// A RenderTarget2D the particle rasteriser blends onto (ParticleSystem.Render draws into whatever target is bound): the native
// lightmap object used as a plain colour target
class RenderTarget {
public:
    RenderTarget(DeviceContext& ctx, int width, int height, int format = ILM_LIGHTMAP_FLOAT4);
    ~RenderTarget();
    RenderTarget(const RenderTarget&) = delete;
    RenderTarget& operator=(const RenderTarget&) = delete;
    void Clear(Vector4 color);
    // float4 texels regardless of the storage format are NOT converted: `dst` receives width * height texels in the target's format
    void Download(void* dst) const;
    IlmHandle Handle() const { return handle; }
    int Width, Height, Format;
private:
    IlmHandle handle = 0;
};

struct SliceInfo {
    int ValidSliceCount = 0;
    std::vector<int> InvalidSlices;
    bool Contains(int index) const;
    void Remove(int index);
};

class DistanceField {
public:
    static constexpr int MaxSurfaceSize = 8192;                    // :19
    static constexpr int DefaultMaximumEncodedDistance = 128;      // :20
    static constexpr int PackedSliceCount = 3;                     // LightingRenderer.PackedSliceCount

    // ctor, :43-122: layout math + the atlas; every slice starts invalid (Invalidate(), :120)
    DistanceField(DeviceContext& ctx, int virtualWidth, int virtualHeight, float virtualDepth, int requestedSliceCount,
                  double requestedResolution = 1, int maximumEncodedDistance = DefaultMaximumEncodedDistance,
                  int format = ILM_SDF_UNORM16);
    virtual ~DistanceField();
    DistanceField(const DistanceField&) = delete;

    int VirtualWidth, VirtualHeight;
    float VirtualDepth;
    double Resolution, RequestedResolution;
    int MaximumEncodedDistance;
    int SliceWidth, SliceHeight, SliceCount, PhysicalSliceCount, ColumnCount, RowCount;
    int TextureWidth, TextureHeight;
    int Format;
    float ZOffset = 0;
    bool NeedClear = true;
    SliceInfo Slices;          // SliceInfo (:41)

    // :56-109 on their own (no device): what the constructor computes before it allocates the atlas
    struct Layout {
        double Resolution = 1;
        int SliceWidth = 0, SliceHeight = 0, SliceCount = 0, PhysicalSliceCount = 0, ColumnCount = 0, RowCount = 0;
        int TextureWidth = 0, TextureHeight = 0;
    };
    static Layout ComputeLayout(int virtualWidth, int virtualHeight, int requestedSliceCount, double requestedResolution = 1);

    // :125-137
    virtual bool IsFullyGenerated() const { return (Slices.ValidSliceCount >= SliceCount) && Slices.InvalidSlices.empty(); }
    virtual bool NeedsRasterize() const { return !Slices.InvalidSlices.empty(); }
    // Save, :183-194: throws InvalidOperationException("The distance field must be fully valid")
    virtual void Save(uint16_t* texels) const;
    // Load, :196-213: raw RGBA16 atlas, 8 bytes per texel; marks every slice valid
    virtual void Load(const uint16_t* texels);
    // :215-231
    virtual void Invalidate();
    virtual void ValidateSlice(int index) { Slices.Remove(index); }
    virtual void MarkValidSlice(int index) { Slices.ValidSliceCount = std::max(Slices.ValidSliceCount, index); }

    IlmHandle Texture() const { return texture; }
    // Uniforms.DistanceField ctor, Uniforms.cs:90-110
    IlmDistanceFieldUniforms GetUniforms() const;
    // what RenderDistanceFieldSliceTriplet binds (LightingRenderer.DistanceField.cs:80-152)
    IlmDistanceFieldRenderDesc GetRenderDesc(int dynamicFlagFilter) const;
protected:
    DeviceContext& context;
    IlmHandle texture = 0;
};

// DynamicDistanceField, SDF/DistanceField.cs:248-321: a static atlas for IsDynamic == false obstructions plus the
// composite one (cleared from the static atlas, dynamic obstructions MAX-blended on top)
class DynamicDistanceField : public DistanceField {
public:
    DynamicDistanceField(DeviceContext& ctx, int virtualWidth, int virtualHeight, float virtualDepth, int sliceCount,
                         double requestedResolution = 1, int maximumEncodedDistance = DefaultMaximumEncodedDistance,
                         int format = ILM_SDF_UNORM16);
    ~DynamicDistanceField() override;
    SliceInfo StaticSliceInfo;
    IlmHandle StaticTexture() const { return staticTexture; }
    void Invalidate() override { Invalidate(true); }
    void Invalidate(bool invalidateStatic);
    void ValidateSlice(int index, bool dynamic);
    void MarkValidSlice(int index, bool dynamic);
    void ValidateSlice(int index) override { ValidateSlice(index, false); ValidateSlice(index, true); }
    void MarkValidSlice(int index) override { MarkValidSlice(index, false); MarkValidSlice(index, true); }
    void Load(const uint16_t*) override { throw std::logic_error("NotImplementedException"); }    // :304-306
    void Save(uint16_t*) const override { throw std::logic_error("NotImplementedException"); }   // :308-310
private:
    IlmHandle staticTexture = 0;
};

namespace Particles {

class ParticleSystem;

// ParticleEngine.cs:616-696
struct ParticleEngineConfiguration {
    int ChunkSize = 256;
    ITimeProvider* TimeProvider = nullptr;
    std::optional<int> UpdatesPerSecond;
    double MaximumUpdateDeltaTimeSeconds = 1.0 / 20;
    bool AccurateLivenessCounts = true;
    explicit ParticleEngineConfiguration(int chunkSize = 256) : ChunkSize(chunkSize) {}
};

// ParticleEngine.cs:24-141
class ParticleEngine {
public:
    static constexpr int RandomnessTextureWidth = 807, RandomnessTextureHeight = 653;   // :45-46
    // The reference fills the randomness texture from an unseeded RNG (:495-544); it is an input here.
    ParticleEngine(DeviceContext& ctx, const ParticleEngineConfiguration& configuration, const float* randomnessTexels);
    ~ParticleEngine();
    ParticleEngine(const ParticleEngine&) = delete;
    DeviceContext& Context;
    ParticleEngineConfiguration Configuration;
    IlmHandle Handle() const { return handle; }
private:
    IlmHandle handle = 0;
};

// Bezier.cs BezierF / Bezier4 reduced to what ClampedBezier1/4 consume (Bezier.cs:442-460, 741-757)
struct BezierF { int Count = 1; int Mode = 0; float MinValue = 0, MaxValue = 1; float A = 1, B = 1, C = 1, D = 1; };
struct Bezier4 { int Count = 1; int Mode = 0; float MinValue = 0, MaxValue = 1; Vector4 A{1, 1, 1, 1}, B{1, 1, 1, 1}, C{1, 1, 1, 1}, D{1, 1, 1, 1}; };
IlmClampedBezier1 MakeClampedBezier1(const std::optional<BezierF>& src);
IlmClampedBezier4 MakeClampedBezier4(const std::optional<Bezier4>& src);

// ParticleConfiguration.cs:13-39
struct ParticleCollision {
    DistanceField* Field = nullptr;                  // DistanceField
    std::optional<float> DistanceFieldMaximumZ;
    float Distance = 0.33f, LifePenalty = 0, EscapeVelocity = 128.0f, BounceVelocityMultiplier = 0.0f;
};
// ParticleConfiguration.cs ParticleColor
struct ParticleColor {
    Vector4 Global{1, 1, 1, 1};                      // :146
    std::optional<float> OpacityFromLife;
    std::optional<Bezier4> ColorFromLife, ColorFromVelocity;
};
// ParticleAppearance (ParticleConfiguration.cs:41-120): the members FillReadbackResult reads; the texture itself lives on the
// C# side, only its size matters here
struct ParticleAppearance {
    std::optional<Vector2> TextureSize;    // Texture.Instance.Width / Height; unset => no texture
    Vector2 OffsetPx{0, 0};
    std::optional<Vector2> SizePx;
    Vector2 AnimationRate{0, 0};
    bool RelativeSize = true;
    bool ColumnFromVelocity = false, RowFromVelocity = false;
    bool Rounded = false, DitheredOpacity = false;                 // :72-77
    bool Bilinear = true;                                          // :87
    BezierF RoundingPowerFromLife{1, 0, 0, 1, 0.8f, 0.8f, 0.8f, 0.8f};   // new BezierF(0.8f), :82
};
// ParticleConfiguration.cs:187-303 (the members the update path reads)
struct ParticleSystemConfiguration {
    ParticleAppearance Appearance;
    bool AutoReadback = false, SortedReadback = false;   // :277-283
    Vector2 Size{1, 1};
    float Friction = 0, MaximumVelocity = 9999.0f, LifeDecayPerSecond = 1;
    std::optional<ParticleCollision> Collision;
    ParticleColor Color;
    std::optional<BezierF> SizeFromLife, SizeFromVelocity;
    float RotationFromLife = 0, RotationFromIndex = 0;   // degrees
    bool RotationFromVelocity = false;
    float ZToY = 0;
    float StippleFactor = 1.0f;                      // :248
    Vector4 ZFormula{0, 0, 0, 0};                    // :282
    float SizeFromZ = 0;                             // :287
    ITimeProvider* TimeProvider = nullptr;
};

namespace Transforms {

enum class AreaType { None = 0, Ellipsoid = 1, Box = 2, Cylinder = 3, Spheroid = 4, Octagon = 5 };
struct TransformArea {   // ParticleTransform.cs TransformArea
    AreaType Type = AreaType::None;
    Vector3 Center, Size{1, 1, 1};
    float Falloff = 1, Rotation = 0;
};

// ParticleTransform.cs:58-292
class ParticleTransform {
public:
    virtual ~ParticleTransform() {}
    bool IsActive = true, IsActive2 = true;
    std::string Label;
    virtual bool IsValid() const = 0;
    virtual bool IsSpawner() const { return false; }
    virtual void Reset() {}
    // the SetParameters half that fills the native op (false => not representable as an op)
    virtual bool FillOp(IlmTransformOp& op, double now) { (void)op; (void)now; return false; }
};

// ParticleTransform.cs:294-326
class ParticleAreaTransform : public ParticleTransform {
public:
    float Strength = 1;
    std::optional<Vector2> CategoryFilter;
    std::optional<TransformArea> Area;
    bool IsValid() const override { return true; }
protected:
    void FillArea(IlmAreaParams& a) const;
};

// Transforms.cs:16-50
class FMA : public ParticleAreaTransform {
public:
    struct FMAParameters { Vector3 Add{0, 0, 0}, Multiply{1, 1, 1}; };
    std::optional<float> CyclesPerSecond = 10.0f;
    FMAParameters Position, Velocity;
    bool FillOp(IlmTransformOp& op, double now) override;
};

// Transforms.cs:133-273
class Noise : public ParticleAreaTransform {
public:
    static constexpr float IntervalUnit = 1000;
    struct P4 { Vector4 Offset{-0.5f, -0.5f, -0.5f, -0.5f}, Minimum{0, 0, 0, 0}, Scale{0, 0, 0, 0}; };
    struct P3 { Vector3 Offset{-0.5f, -0.5f, -0.5f}, Minimum{0, 0, 0}, Scale{1, 1, 1}; };
    struct PF { float Offset = -0.5f, Minimum = 0, Scale = 0; };
    std::optional<float> CyclesPerSecond = 10.0f;
    P4 Position;
    P3 Velocity;
    PF Speed;
    float Interval = IntervalUnit;   // milliseconds
    bool ReplaceOldVelocity = true;
    explicit Noise(uint64_t seed = 1);
    void Reset() override;
    bool FillOp(IlmTransformOp& op, double now) override;
    double CurrentU = 0, CurrentV = 0, NextU = 0, NextV = 0;
private:
    void CycleUVs();
    void AutoCycleUV(float now, double intervalSecs, float& t);
    double LastUChangeWhen = 0;
    Xoshiro RNG;
};

// Transforms.cs:275-372
enum class AttractorType { Physical = 0, Linear = 1, Exponential = 2 };
class Gravity : public ParticleTransform {
public:
    static constexpr int MaxAttractors = 16;
    struct Attractor { Vector3 Position; float Radius = 1, Strength = 1; AttractorType Type = AttractorType::Linear; };
    float MaximumAcceleration = 8;
    std::vector<Attractor> Attractors;
    // The reference never binds CategoryFilter for Gravity (it derives from ParticleTransform): the effect default
    // (0, 0) applies, i.e. only category-0 particles are attracted.  Kept explicit so the quirk is visible.
    Vector2 CategoryFilter{0, 0};
    bool IsValid() const override { return !Attractors.empty(); }
    bool FillOp(IlmTransformOp& op, double now) override;
};

enum class FormulaType { Linear = 0, Spherical = 1, Towards = 2, Rectangular = 3 };
struct Formula1 { float Constant = 0, RandomScale = 0, Offset = 0; };
struct Formula3 {
    Vector3 Constant, RandomScale, Offset;
    FormulaType Type = FormulaType::Linear;
    bool Circular() const { return Type == FormulaType::Spherical || Type == FormulaType::Rectangular; }
    static Formula3 UnitNormal() { Formula3 f; f.RandomScale = {1, 1, 1}; f.Type = FormulaType::Spherical; return f; }   // Formula.cs UnitNormal
};
struct Formula4 { Vector4 Constant{1, 1, 1, 1}, RandomScale, Offset; };

// ParticleSpawner.cs:16-260
class SpawnerBase : public ParticleTransform {
public:
    float MinRate = 0, MaxRate = 0;
    std::optional<int> MaximumTotal;
    Formula3 Position = Formula3::UnitNormal();
    IlmMatrix PositionPostMatrix, VelocityPostMatrix;
    bool AlignVelocityAndPosition = false;
    Vector3 AxisMask{1, 1, 1};
    Formula3 Velocity = Formula3::UnitNormal();
    Formula1 Life{1, 0, 0};
    Formula1 Category{0, 0, 0};
    Formula4 Color;
    float AlphaDiscardThreshold = 1;

    explicit SpawnerBase(uint64_t seed = 1);
    bool IsSpawner() const override { return true; }
    bool IsValid() const override { return true; }
    void Reset() override { RateError = 0; totalSpawned = 0; }
    int TotalSpawned() const { return totalSpawned; }
    double RateError = 0;
    // The reference draws the rate from an unseeded RNG (ParticleSpawner.cs:165); draws queued here are consumed
    // first, in order, so a test can replay the exact sequence a fixture was derived for.
    std::vector<double> ScriptedDraws;

    virtual bool PartialSpawnAllowed() const { return true; }
    virtual int CountScale() const { return 1; }
    // :148-150
    virtual double AdjustCurrentRate(double rate) { return rate; }
    // :152-189
    virtual void BeginTick(double now, double deltaTimeSeconds, int& spawnCount);
    // BeginTick as RunSpawner calls it (with the target system and the source chunk out-parameter, :152);
    // sourceChunkIndex = index in the SOURCE system's chunk table, -1 when there is none
    virtual void BeginTick(ParticleSystem& system, double now, double deltaTimeSeconds, int& spawnCount, int& sourceChunkIndex) {
        (void)system; sourceChunkIndex = -1; BeginTick(now, deltaTimeSeconds, spawnCount);
    }
    // AddError, :196-198
    void AddError(double amount) { RateError += amount; }
    // fills the whole record (kind, params, feedback block); positions receives the PositionBuffer contents when the
    // record is of kind ILM_SPAWN_POSITION_BUFFER
    virtual void FillRecord(IlmSpawnRecord& rec, std::vector<IlmFloat4>& positions, int chunkSize, double now) {
        positions.clear(); rec.Kind = ILM_SPAWN_INLINE; FillSpawn(rec.Params, chunkSize, now);
    }
    virtual bool IsFeedback() const { return false; }
    // device-side resources a record of this spawner needs in spawn record slot `slot` of `system` (the pattern texture)
    virtual void BindResources(ParticleSystem& system, int slot) { (void)system; (void)slot; }
    // RunSpawner's consumed-count bookkeeping for feedback sources (ParticleSpawning.cs:159-166)
    virtual void OnSpawned(int spawnCount) { (void)spawnCount; }
    // :191-194
    void EndTick(int requestedSpawnCount, int actualSpawnCount);
    void SetIndices(int first, int last) { indexFirst = first; indexLast = last; }
    // :200-256
    virtual void FillSpawn(IlmSpawnParams& p, int chunkSize, double now);
    // :142-150
    float EstimateMaximumLifeForNewParticle() const;
protected:
    double NextRateDraw();
    int indexFirst = 0, indexLast = 0, totalSpawned = 0;
    Xoshiro RNG;
};

// ParticleSpawner.cs:262-419
class Spawner : public SpawnerBase {
public:
    static constexpr int MaxInlinePositions = 4;
    std::vector<Vector3> AdditionalPositions;
    std::optional<float> PolygonRate;
    bool PolygonLoop = true;
    Formula1 VelocityAlongPolygon{0, 0, 0};
    bool RatePerPosition = true;
    explicit Spawner(uint64_t seed = 1) : SpawnerBase(seed) {}
    int CountScale() const override;
    void FillSpawn(IlmSpawnParams& p, int chunkSize, double now) override;
    // more than MaxInlinePositions positions => technique SpawnParticlesFromPositionTexture (GetMaterial, :295-299)
    void FillRecord(IlmSpawnRecord& rec, std::vector<IlmFloat4>& positions, int chunkSize, double now) override;
};

// SpecialSpawners.cs:243-443
class FeedbackSpawner : public SpawnerBase {
public:
    explicit FeedbackSpawner(uint64_t seed = 1) : SpawnerBase(seed) {}
    ParticleSystem* SourceSystem = nullptr;       // ParticleSystemReference, resolved
    std::optional<int> SlidingWindowSize;
    int SlidingWindowMargin = 0;
    bool SpawnFromEntireWindow = false;
    int InstanceMultiplier = 1;
    bool AlignPositionConstant = true;
    float SourceVelocityFactor = 0.0f;
    bool MultiplyLife = false, MultiplyColorConstant = false;
    Vector2 SourceLifeRange{0, 9999};
    bool IsFeedback() const override { return true; }
    void Reset() override { SpawnerBase::Reset(); currentFeedbackSourceIndex = 0; currentFeedbackSource = -1; }
    void BeginTick(ParticleSystem& system, double now, double deltaTimeSeconds, int& spawnCount, int& sourceChunkIndex) override;
    void FillRecord(IlmSpawnRecord& rec, std::vector<IlmFloat4>& positions, int chunkSize, double now) override;
    void OnSpawned(int spawnCount) override;
private:
    int currentFeedbackSource = -1;        // chunk table index in the source system
    int currentFeedbackSourceIndex = 0;
};

// SpecialSpawners.cs:15-264.  Texture = the mip levels as float4 texels (the reference's Texture2D comes from its texture loader).
class PatternSpawner : public SpawnerBase {
public:
    explicit PatternSpawner(uint64_t seed = 1) : SpawnerBase(seed) {}
    std::optional<Vector2> TextureTopLeftPx, TextureSizePx;
    float MipBiasBase = -0.5f;
    bool WholeSpawn = false, InstantInitialSpawn = true, MultiplyColorConstant = true;
    int Divisor() const { return divisor; }
    void SetDivisor(int v) { divisor = std::min(std::max(v, 1), 10); }          // Arithmetic.Clamp(value, 1, 10), :44-51
    // `texels`: `levels` mip levels back to back, level l = max(1, width >> l) x max(1, height >> l)
    void SetTexture(int width, int height, int levels, std::vector<IlmFloat4> texels);
    int TextureWidth() const { return texWidth; }
    int TextureHeight() const { return texHeight; }
    int ParticlesPerRow() const;       // :111-115
    int RowsPerInstance() const;       // :117-121
    int ParticlesPerInstance() const { return ParticlesPerRow() * RowsPerInstance(); }
    int RowsSpawned() const { return rowsSpawned; }
    int CountScale() const override { return WholeSpawn ? ParticlesPerInstance() : ParticlesPerRow(); }
    bool PartialSpawnAllowed() const override { return false; }
    bool IsValid() const override { return texLevels > 0; }
    void Reset() override { SpawnerBase::Reset(); rowsSpawned = 0; }
    double AdjustCurrentRate(double rate) override;
    void BeginTick(ParticleSystem& system, double now, double deltaTimeSeconds, int& spawnCount, int& sourceChunkIndex) override;
    void FillRecord(IlmSpawnRecord& rec, std::vector<IlmFloat4>& positions, int chunkSize, double now) override;
    void BindResources(ParticleSystem& system, int slot) override;
private:
    Vector2 DirectTextureSize() const;  // :76-97
    int divisor = 1, rowsSpawned = 0;
    int texWidth = 0, texHeight = 0, texLevels = 0;
    uint64_t texVersion = 0;
    std::vector<IlmFloat4> texData;
};

// Transforms.cs:52-71
class MatrixMultiply : public ParticleAreaTransform {
public:
    std::optional<float> CyclesPerSecond = 10.0f;
    IlmMatrix Position, Velocity;
    MatrixMultiply();
    bool FillOp(IlmTransformOp& op, double now) override;
};

// Transforms.cs:275-300
class SpatialNoise : public Noise {
public:
    Vector2 SpaceScale{1, 1};
    explicit SpatialNoise(uint64_t seed = 1) : Noise(seed) {}
    bool FillOp(IlmTransformOp& op, double now) override;
};

}  // namespace Transforms

// ParticleSystem.cs:48-1072 (update path), ParticleSpawning.cs, ParticleLiveness.cs
class ParticleSystem {
public:
    static constexpr int MaxChunkCount = 64;             // ParticleSystem.cs:49
    static constexpr int LivenessCheckInterval = 4;      // ParticleLiveness.cs:14
    int DeadFrameThreshold = LivenessCheckInterval * 4;  // ParticleLiveness.cs:22
    // true: wait for the liveness counts at the next Update (deterministic, for tests); false: poll them without
    // stalling the stream, as the reference's LivenessDataReadbackWorkItem does (ParticleWorkItems.cs:98-135)
    bool BlockingLivenessReadback = false;

    struct Chunk {   // ParticleSystem.cs:148-240
        int ID = 0;
        int NextSpawnOffset = 0, TotalSpawned = 0;
        bool NoLongerASpawnTarget = false;
        bool IsFeedbackSource = false;                     // ParticleSystem.cs:164
        int TotalConsumedForFeedback = 0;                  // :168
        int AvailableForFeedback() const { return TotalSpawned - TotalConsumedForFeedback; }   // :170-174
        int FeedbackSourceIndex() const { return TotalConsumedForFeedback; }                   // :175-179
        void SkipFeedbackInput(int skipAmount) {           // :235-239
            TotalConsumedForFeedback += skipAmount;
            if (TotalConsumedForFeedback > TotalSpawned) TotalConsumedForFeedback = TotalSpawned;
        }
        float ApproximateMaximumLife = 0;
        // LivenessInfo, ParticleLiveness.cs:24-28
        std::optional<int> Count;
        int DeadFrameCount = 0;
    };
    // UpdateResult, ParticleSystem.cs:51-71
    struct UpdateResult { bool PerformedUpdate = false; float Timestamp = 0; };

    ParticleSystem(ParticleEngine& engine, const ParticleSystemConfiguration& configuration);
    ~ParticleSystem();
    ParticleSystem(const ParticleSystem&) = delete;

    ParticleEngine& Engine;
    ParticleSystemConfiguration Configuration;
    std::vector<Transforms::ParticleTransform*> Transforms;   // not owned
    int LiveCount = 0;
    int Capacity() const { return (int)chunks.size() * ChunkMaximumCount(); }
    int ChunkMaximumCount() const { return Engine.Configuration.ChunkSize * Engine.Configuration.ChunkSize; }
    const std::vector<Chunk>& Chunks() const { return chunks; }
    long TotalSpawnCount = 0;

    // Spawn(particleCount, initializers), ParticleSpawning.cs:61-113: whole new chunks filled from host arrays
    // (position/velocity/color are float4 arrays of particleCount entries; color may be null => zeros).
    int Spawn(int particleCount, const IlmFloat4* positions, const IlmFloat4* velocities, const IlmFloat4* colors);
    // Update, ParticleSystem.cs:630-761.  frameIndex plays DeviceManager.FrameIndex.
    UpdateResult Update(int frameIndex);
    void Clear() { isClearPending = true; }   // ParticleSystem.cs:1000-1003
    // AutoReadback / ReadbackResult (ParticleReadback.cs:21-71) reduced to a synchronous plane download
    void Readback(int chunkIndex, int plane, IlmFloat4* dst) const;
    // MaybePerformReadback + FillReadbackResult (ParticleReadback.cs:21-167): one draw-call record per live particle, chunk / slot
    // order.  Update() refreshes ReadbackResult when Configuration.AutoReadback is set (the reference completes a Future).
    std::vector<IlmReadbackDrawCall> PerformReadback() const;
    // the same records where the native layer left them (page-locked host memory owned by the device context; valid until the
    // next read-back on that context): ArraySegment<BitmapDrawCall> over the pooled buffer, ParticleReadback.cs:40-41,64-69
    struct ReadbackView { const IlmReadbackDrawCall* Records = nullptr; int Count = 0; };
    ReadbackView PerformReadbackView() const;
    IlmReadbackParams GetReadbackParams() const;
    // ParticleRenderParameters, ParticleConfiguration.cs:305-312
    struct RenderParameters { Vector2 Origin{0, 0}, Scale{1, 1}; std::optional<float> StippleFactor; };
    struct RenderStats { uint64_t LiveQuads = 0, TilePairs = 0, ShadedPixels = 0; };
    // Appearance.Texture.Instance for the textured techniques: width x height float4 texels, one level (the native layer binds
    // bitmaps without mips only); Appearance.TextureSize must say the same size
    void SetBitmap(int width, int height, const IlmFloat4* texels);
    // ParticleSystem.Render (ParticleSystem.cs:943-1041): technique RasterizeParticlesNoTexture, or TextureLinear / TexturePoint by
    // Appearance.Bilinear when a texture is set (:963-971), every chunk in order with quadCount = min(ChunkMaximumCount, TotalSpawned + 1) (:880), blended onto
    // `target` with blendMode (ILM_BLEND_*); viewportScale / viewportPosition play Fracture's ViewTransform.
    RenderStats Render(RenderTarget& target, int blendMode = ILM_BLEND_ALPHA, const RenderParameters* renderParams = nullptr,
                       Vector2 viewportScale = Vector2{1, 1}, Vector2 viewportPosition = Vector2{0, 0}, bool wantStats = false) const;
    // Uniforms.RasterizeParticleSystem + RenderingOptions + StippleFactor as Render binds them (Uniforms.cs:238-290, ParticleSystem.cs:254-271,1023-1032)
    IlmRasterizeParams GetRasterizeParams(int blendMode, const RenderParameters* renderParams, Vector2 viewportScale, Vector2 viewportPosition) const;
    std::vector<IlmReadbackDrawCall> ReadbackResult;
    float ReadbackTimestamp = 0;
    IlmHandle Handle() const { return handle; }
    // PickSourceForFeedback / GetCurrentSpawnTarget, ParticleSpawning.cs:233-265 (chunk table indices, -1 = null)
    int PickSourceForFeedback(int count);
    int GetCurrentSpawnTarget(bool feedback) const;
    Chunk& ChunkAt(int index) { return chunks.at((size_t)index); }
    // the descriptor of the last launch (tests compare it with the oracle's step)
    const IlmStepDesc& LastStep() const { return lastStep; }
    double LastDeltaTimeSeconds = 0;
    // which pattern texture (owner, version) spawn record slot `slot` currently holds on the device; true = already bound
    bool PatternBound(int slot, const void* owner, uint64_t version) {
        if (boundPatternOwner[slot] == owner && boundPatternVersion[slot] == version) return true;
        boundPatternOwner[slot] = owner; boundPatternVersion[slot] = version;
        return false;
    }

private:
    bool RunSpawner(Transforms::SpawnerBase& spawner, double deltaTimeSeconds, double now, bool isSecondPass,
                    std::vector<IlmSpawnRecord>& records, std::vector<std::vector<IlmFloat4>>& recordPositions,
                    std::vector<Transforms::SpawnerBase*>& recordSpawners);
    int PickTargetForSpawn(bool feedback, int count, bool& needClear, bool partialSpawnAllowed);
    int CreateChunk();
    void UpdateLiveCountAndReapDeadChunks();
    void ProcessLatestLivenessInfo(Chunk& c);
    void FillSystemUniforms(IlmStepDesc& d, double deltaTimeSeconds) const;
    void Launch(const IlmStepDesc& d);

    IlmHandle handle = 0;
    std::vector<Chunk> chunks;
    std::vector<int> chunksToReap;   // chunk IDs
    int nextChunkId = 1;
    int currentSpawnTarget = -1, currentFeedbackSpawnTarget = -1, currentFeedbackSource = -1;   // chunk IDs
    int currentFrameIndex = 0;
    int lastFrameUpdated = -1;
    int framesUntilNextLivenessCheck = 0;
    bool livenessPending = false;
    std::vector<int> livenessChunkIds;   // chunk IDs in table order when the pending count was issued
    std::optional<double> lastUpdateTimeSeconds;
    double updateErrorAccumulator = 0;
    bool isClearPending = false;
    IlmStepDesc lastStep;
    ManualTimeProvider defaultTime;
    const void* boundPatternOwner[ILM_MAX_SPAWNS] = {};
    uint64_t boundPatternVersion[ILM_MAX_SPAWNS] = {};
};

}  // namespace Particles

namespace Lighting {

enum class LightSourceRampMode { Linear = 0, Exponential = 1, Constant = 2 };       // LightSource.cs
enum class LightShadowFilter { None = -1, ShadowsOnly = 1, NoShadowsOnly = 0 };   // as packed into EvenMoreLightProperties.x

// LightSource.cs:37-280 (SphereLightSource)
// A light ramp texture (LightSource.TextureRef / RendererConfiguration.DefaultRampTexture): width x height float4 texels.  Lights are
// grouped by the texture they share (LightTypeRenderStateKey.RampTexture, LightingRenderer.cs:50-83); a 1 x 1 ramp is no ramp (:822-827).
struct RampTexture {
    int Width = 0, Height = 0;
    std::vector<IlmFloat4> Texels;
};

struct RendererQualitySettings;

struct SphereLightSource {
    int SortKey = 0;          // LightSourceBase.SortKey: RenderLighting sorts by it first (LightSorter, LightingRenderer.cs:2066-2096)
    bool Enabled = true;      // LightSourceBase.Enabled (LightSource.cs:42): disabled lights are skipped (LightingRenderer.cs:1062)
    Vector3 Position;
    float Radius = 0, RampLength = 1;
    Vector4 Color{1, 1, 1, 1};
    float Opacity = 1;
    LightSourceRampMode RampMode = LightSourceRampMode::Linear;
    bool CastsShadows = true;
    float AmbientOcclusionRadius = 0, AmbientOcclusionOpacity = 1;
    std::optional<float> ShadowDistanceFalloff;
    float FalloffYFactor = 1;
    int ShadowFilter = -1;
    Vector3 SpecularColor{0, 0, 0};
    float SpecularPower = 1;
    std::shared_ptr<RampTexture> TextureRef;      // LightSource.TextureRef; null => Configuration.DefaultRampTexture
    std::shared_ptr<RendererQualitySettings> Quality;   // LightSource.Quality (LightSource.cs:95); null => Configuration.DefaultQuality
    float RampOffset = 0, RampRate = 1;           // RampOffsetAndRate, LightSource.cs:90
};

// ReplicatedLight / LightSourceReplicator, LightSource.cs:601-620: one template, many placements.  Each placement may override the
// per-light values; everything else (ramp mode, shadows, AO, falloff, ramp texture, quality) comes from the template.
struct ReplicatedLight {
    Vector3 Position;
    std::optional<float> Radius, RampLength, SpecularPower, Opacity;
    std::optional<Vector4> Color;
    std::optional<Vector3> SpecularColor;
};
struct LightSourceReplicator {
    int SortKey = 0;
    bool Enabled = true;
    SphereLightSource Template;
    std::vector<ReplicatedLight> Lights;
    void Clear() { Lights.clear(); }
    void Add(const ReplicatedLight& l) { Lights.push_back(l); }
};

// ParticleLightSource, LightSource.cs:466-505
struct ParticleLightSource {
    SphereLightSource Template;
    Particles::ParticleSystem* System = nullptr;
    bool IsActive = true, Enabled = true;
    std::optional<float> StippleFactor;     // defaults to the system's (1): only >= 1 is supported (StippleReject is Fracture code)
};

// LightProbe / LightProbeCollection, LightProbe.cs:15-160
struct LightProbe {
    Vector3 Position;
    std::optional<Vector3> Normal;
    bool EnableShadows = true;
    Vector4 PreviousValue, Value;
};
class LightProbeCollection {
public:
    explicit LightProbeCollection(int maximumCount) : MaximumCount(maximumCount) {}
    const int MaximumCount;
    bool IsDirty = false;
    std::vector<std::shared_ptr<LightProbe>> Items;
    void Add(std::shared_ptr<LightProbe> probe) {
        if ((int)Items.size() >= MaximumCount) throw InvalidOperationException("List full");    // :32-33
        Items.push_back(std::move(probe)); IsDirty = true;
    }
    void Clear() { Items.clear(); IsDirty = true; }
    int Count() const { return (int)Items.size(); }
};

// LightObstruction.cs:10-140
enum class LightObstructionType : short { Ellipsoid = 0, Box = 1, Cylinder = 2, Spheroid = 3, Octagon = 4 };
class LightObstruction {
public:
    LightObstruction(LightObstructionType type, Vector3 center = {}, Vector3 radius = {}, float rotation = 0);
    static LightObstruction Box(Vector3 center, Vector3 size, float rotation = 0) { return LightObstruction(LightObstructionType::Box, center, size, rotation); }
    static LightObstruction Ellipsoid(Vector3 center, Vector3 size, float rotation = 0) { return LightObstruction(LightObstructionType::Ellipsoid, center, size, rotation); }
    static LightObstruction Cylinder(Vector3 center, Vector3 size, float rotation = 0) { return LightObstruction(LightObstructionType::Cylinder, center, size, rotation); }
    // every setter invalidates when the value changes (:38-91)
    LightObstructionType Type() const { return type; }
    void SetType(LightObstructionType v) { if (type != v) Invalidate(); type = v; }
    Vector3 Center() const { return center; }
    void SetCenter(Vector3 v) { if (v.X != center.X || v.Y != center.Y || v.Z != center.Z) Invalidate(); center = v; }
    Vector3 Size() const { return size; }
    void SetSize(Vector3 v) { if (v.X != size.X || v.Y != size.Y || v.Z != size.Z) Invalidate(); size = v; }
    Vector4 Orientation() const { return orientation; }
    void SetOrientation(Vector4 q) { if (q.X != orientation.X || q.Y != orientation.Y || q.Z != orientation.Z || q.W != orientation.W) Invalidate(); orientation = q; }
    // Rotation setter, :94-103: Quaternion.CreateFromAxisAngle(Vector3.UnitZ, value)
    std::optional<float> Rotation() const { return shadowRotation; }
    void SetRotation(float value);
    bool IsDynamic() const { return isDynamic; }
    void SetIsDynamic(bool v) { if (isDynamic != v) HasDynamicityChanged = true; isDynamic = v; }
    void Invalidate() { IsValid = false; }
    bool IsValid = false, HasDynamicityChanged = true;
    IlmObstruction Vertex() const;   // DistanceFunctionVertex (Vertices.cs:105-141) + type + dynamic flag
private:
    LightObstructionType type;
    Vector3 center, size;
    Vector4 orientation{0, 0, 0, 1};
    std::optional<float> shadowRotation;
    bool isDynamic = false;
};

// LightObstructionCollection, LightingEnvironment.cs:51-135
class LightObstructionCollection {
public:
    bool IsInvalid = true, IsInvalidDynamic = true;
    std::vector<std::shared_ptr<LightObstruction>> Items;
    void Add(std::shared_ptr<LightObstruction> value) { (value->IsDynamic() ? IsInvalidDynamic : IsInvalid) = true; Items.push_back(std::move(value)); }
    void RemoveAt(int index) { (Items.at(index)->IsDynamic() ? IsInvalidDynamic : IsInvalid) = true; Items.erase(Items.begin() + index); }
    void Clear() { IsInvalid = true; Items.clear(); }
    int Count() const { return (int)Items.size(); }
};

// HeightVolumeBase / SimpleHeightVolume, SDF/HeightVolume.cs:14-80 (the members the distance field pass reads)
struct HeightVolume {
    std::vector<Vector2> Polygon;
    float ZBase = 0, Height = 0;
    bool IsDynamic = true;      // :23
    bool IsObstruction = true;
    bool TopFaceEnableShadows = true;   // :18
};

// LightingEnvironment.cs:13-49
struct LightingEnvironment {
    std::vector<SphereLightSource> Lights;
    std::vector<LightSourceReplicator> Replicators;    // LightSourceReplicator entries of Lights in the reference
    std::vector<ParticleLightSource> ParticleLights;   // ParticleLightSource entries of Lights in the reference (one render state each)
    LightObstructionCollection Obstructions;
    std::vector<HeightVolume> HeightVolumes;
    float GroundZ = 0, MaximumZ = 128, ZToYMultiplier = 0;
    Vector4 Ambient{0, 0, 0, 1};
    bool EnableGroundShadows = true;    // :41
};

// LightingRenderer.Configuration.cs:254-291
struct RendererQualitySettings {
    float MinStepSize = 3.0f, LongStepFactor = 1.0f;
    int MaxStepCount = 64;
    float MaxConeRadius = 24, ConeGrowthFactor = 1.0f, OcclusionToOpacityPower = 1;
};

// LightingRenderer.Configuration.cs:13-250 (members the sphere-light pass reads)
struct RendererConfiguration {
    int RenderWidth, RenderHeight;
    bool HighQuality = true;          // HalfVector4 lightmap (:206); false => Color
    bool TwoPointFiveD = false;
    float LightOcclusion = 0;
    Vector2 RenderScale{1, 1};
    RendererQualitySettings DefaultQuality;
    int MaximumFieldUpdatesPerFrame = 1;   // LightingRenderer.Configuration.cs:91
    int MaximumLightProbeCount = 256;      // LightingRenderer.Configuration.cs
    bool EnableGBuffer = false;            // the mirror starts without one (SetGBuffer uploads, UpdateFields generates)
    bool RenderGroundPlane = true;
    bool HighQualityGBuffer = true;        // GBuffer format Vector4 (true) or HalfVector4, GBuffer.cs:30-38
    bool FloatLightmap = false;       // extension: fp32 lightmap (parity format)
    std::shared_ptr<RampTexture> DefaultRampTexture;   // LightingRenderer.Configuration.cs:78
    RendererConfiguration(int w, int h) : RenderWidth(w), RenderHeight(h) {}
};

// LightingRenderer.cs (RenderLighting path only)
class LightingRenderer {
public:
    // externalLightmap: optional caller-owned device memory for the lightmap (e.g. a tensor RCCL all-gathers)
    LightingRenderer(DeviceContext& ctx, const RendererConfiguration& configuration, LightingEnvironment* environment,
                     void* externalLightmap = nullptr);
    ~LightingRenderer();
    LightingRenderer(const LightingRenderer&) = delete;

    DeviceContext& Context;
    RendererConfiguration Configuration;
    LightingEnvironment* Environment;
    DistanceField* Field = nullptr;       // DistanceField property, :594-607
    LightProbeCollection Probes;          // :330
    // G-buffer: null => ground plane only
    void SetGBuffer(const void* texels, int width, int height, int format);

    // InvalidateFields, :1942-1947
    void InvalidateFields(bool invalidateDistanceField = true) { if (invalidateDistanceField && Field) Field->Invalidate(); }
    // UpdateFields, :1949-1975 (distance field half): AutoInvalidateDistanceField (:1977-2014), then at most
    // MaximumFieldUpdatesPerFrame slices of RenderDistanceField (LightingRenderer.DistanceField.cs:20-30,415-464).
    // Returns the number of slice triplets rendered.
    int UpdateFields();
    // RenderGBuffer, LightingRenderer.GBuffer.cs:127-219 (non-2.5D: ground plane + height-volume top faces); called by UpdateFields
    // when Configuration.EnableGBuffer is set
    void RenderGBuffer(Vector2 viewportPosition = {0, 0}, Vector2 viewportScale = {1, 1});
    IlmHandle GBuffer() const { return gbuffer; }

    // RenderLighting, :917-1191: clears to Ambient * intensityScale and adds every sphere light.
    // [rowBegin, rowEnd) restricts the pass to a screen strip (multi-GPU split); rowEnd < 0 => whole frame.
    void RenderLighting(float intensityScale = 1.0f, int rowBegin = 0, int rowEnd = -1, IlmRenderStats* stats = nullptr);
    void ReadLightmap(void* dst, int firstRow, int rowCount) const;
    // RenderedLighting.Resolve -> ResolveLighting without albedo (LightingRenderer.HDR.cs:99-151, LightingRenderer.cs:1537-1645):
    // tone-maps the lightmap into `destination` (a lightmap handle of the same size; RGBA8 = the back buffer).  hdr == nullptr => the
    // plain LightingResolve with exposure / gamma 1.
    // `albedo` != 0: the ...WithAlbedo techniques over a texture of the lightmap's size (ResolveLighting with albedo != null, :1549-1580).
    void Resolve(IlmHandle destination, const IlmHDRConfiguration* hdr = nullptr, int rowBegin = 0, int rowEnd = -1, IlmHandle albedo = 0) const;
    IlmHandle Lightmap() const { return lightmap; }
    int LightmapFormat() const { return lightmapFormat; }

    // _ParticleLightBatchSetup, :769-790
    static IlmParticleLightParams PackParticleLight(const ParticleLightSource& pls, bool haveDistanceField);
    // RenderSphereLightSource, :1193-1219
    static bool PackSphereLight(const SphereLightSource& l, float intensityScale, bool haveDistanceField, IlmLightVertex& v);
    // SetDistanceFieldParameters, :1894-1940
    IlmDistanceFieldUniforms GetDistanceFieldUniforms(const RendererQualitySettings& q) const;
    // ComputeUniforms :691-701 + SetGBufferParameters LightingRenderer.GBuffer.cs:520-534
    IlmEnvironment GetEnvironmentUniforms() const;
    // the LightVertex stream of the last RenderLighting, in draw order (LightTypeRenderState.LightVertices)
    const std::vector<IlmLightVertex>& PackedLightVertices() const { return vertices; }

private:
    void UpdateLightProbes(float intensityScale);   // LightingRenderer.LightProbes.cs:49-150 (synchronous read-back)
    void AutoInvalidateDistanceField();
    int RenderDistanceFieldPartition(int dynamicFlagFilter /* -1 = null */);
    IlmHandle lightmap = 0, gbuffer = 0;
    int lightmapFormat = ILM_LIGHTMAP_HALF4;
    int gbufferWidth = 0, gbufferHeight = 0;
    std::vector<IlmLightVertex> vertices;
    // the light groups of the last RenderLighting (one per ramp texture) and the ramp currently bound on the context
    std::vector<const RampTexture*> groupKeys;
    std::vector<const RendererQualitySettings*> groupQuality;     // null => Configuration.DefaultQuality
    std::vector<std::vector<IlmLightVertex>> groups;
    const RampTexture* boundRamp = nullptr;
    void BindRamp(const RampTexture* ramp);
};

}  // namespace Lighting
}  // namespace Illuminant
}  // namespace Squared

// illuminant_host.cpp -- host-side mirror of the reference's C# logic above the C ABI (see the header).
#include "illuminant_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace Squared {
namespace Illuminant {

void ThrowIfFailed(int32_t code) {
    if (code == ILM_OK)
        return;
    const char* msg = ilm_last_error();
    throw NativeException(code, std::string("illuminant_hip error ") + std::to_string(code) + ": " + (msg ? msg : ""));
}

// ---- Xoshiro --------------------------------------------------------------------------------------------
static uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
Xoshiro::Xoshiro(uint64_t seed) {
    for (int i = 0; i < 4; i++) s[i] = splitmix64(seed);
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t Xoshiro::NextUInt64() {
    const uint64_t result = rotl(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
}
double Xoshiro::NextDouble() { return (double)(NextUInt64() >> 11) * (1.0 / 9007199254740992.0); }

// ---- DeviceContext ---------------------------------------------------------------------------------------
DeviceContext::DeviceContext(int deviceId) { ThrowIfFailed(ilm_ctx_create(deviceId, &handle)); }
DeviceContext::~DeviceContext() { if (handle && owned) ilm_ctx_destroy(handle); }
void DeviceContext::Sync() { ThrowIfFailed(ilm_ctx_sync(handle)); }
void DeviceContext::TimerStart() { ThrowIfFailed(ilm_timer_start(handle)); }
float DeviceContext::TimerStop() { float ms = 0; ThrowIfFailed(ilm_timer_stop(handle, &ms)); return ms; }

RenderTarget::RenderTarget(DeviceContext& ctx, int width, int height, int format) : Width(width), Height(height), Format(format) {
    ThrowIfFailed(ilm_lightmap_create(ctx.Handle(), width, height, format, nullptr, &handle));
}
RenderTarget::~RenderTarget() { if (handle) ilm_lightmap_destroy(handle); }
void RenderTarget::Clear(Vector4 color) {
    const float c[4] = { color.X, color.Y, color.Z, color.W };
    ThrowIfFailed(ilm_lightmap_clear(handle, c));
}
void RenderTarget::Download(void* dst) const { ThrowIfFailed(ilm_lightmap_download(handle, dst, 0, Height)); }

// ---- DistanceField, SDF/DistanceField.cs:43-122 -------------------------------------------------------------
static double RoundToEven(double v) { return std::nearbyint(v); }   // Math.Round: MidpointRounding.ToEven

// The layout half of the constructor (:56-109): pure integer / double arithmetic, no device.
DistanceField::Layout DistanceField::ComputeLayout(int virtualWidth, int virtualHeight, int requestedSliceCount, double requestedResolution) {
    Layout L;
    if (requestedResolution < 0.05) requestedResolution = 0.05;
    else if (requestedResolution > 1) requestedResolution = 1;

    const int candidateSliceWidth = (int)RoundToEven(virtualWidth * requestedResolution);
    const int candidateSliceHeight = (int)RoundToEven(virtualHeight * requestedResolution);
    const double fracX = (double)virtualWidth / candidateSliceWidth, fracY = (double)virtualHeight / candidateSliceHeight;
    const double frac = (fracX + fracY) / 2;
    double resolution = RoundToEven((1.0 / frac) * 1000.0) / 1000.0;   // Math.Round(x, 3)
    if (resolution < 0.05) resolution = 0.05;
    else if (resolution > 1) resolution = 1;
    L.Resolution = resolution;

    L.SliceWidth = (int)RoundToEven(virtualWidth * L.Resolution);
    L.SliceHeight = (int)RoundToEven(virtualHeight * L.Resolution);
    const int maxSlicesX = MaxSurfaceSize / L.SliceWidth, maxSlicesY = MaxSurfaceSize / L.SliceHeight;
    const int maxSlices = maxSlicesX * maxSlicesY * PackedSliceCount;

    int sliceCount = std::max(3, requestedSliceCount);
    sliceCount = ((sliceCount + 2) / 3) * 3;
    L.SliceCount = std::min(sliceCount, maxSlices);
    L.PhysicalSliceCount = (int)std::ceil(L.SliceCount / (float)PackedSliceCount);

    L.ColumnCount = std::min(maxSlicesX, L.PhysicalSliceCount);
    L.RowCount = std::min(maxSlicesY, std::max((int)std::ceil(L.PhysicalSliceCount / (float)maxSlicesX), 1));
    // "HACK: If the DF is going to be extremely wide but not tall, rebalance it" (:91-109)
    while ((L.RowCount < L.ColumnCount) && (L.RowCount < maxSlicesY)) {
        int newRowCount = L.RowCount + 1;
        int newColumnCount = (int)std::ceil(L.PhysicalSliceCount / (float)newRowCount);
        if (newRowCount > maxSlicesX) newRowCount = maxSlicesX;
        if (newColumnCount > maxSlicesY) newColumnCount = maxSlicesY;
        if ((newRowCount * newColumnCount) < L.PhysicalSliceCount) break;
        L.RowCount = newRowCount;
        L.ColumnCount = newColumnCount;
    }
    L.TextureWidth = L.SliceWidth * L.ColumnCount;
    L.TextureHeight = L.SliceHeight * L.RowCount;
    return L;
}

bool SliceInfo::Contains(int index) const { return std::find(InvalidSlices.begin(), InvalidSlices.end(), index) != InvalidSlices.end(); }
void SliceInfo::Remove(int index) {
    auto it = std::find(InvalidSlices.begin(), InvalidSlices.end(), index);
    if (it != InvalidSlices.end()) InvalidSlices.erase(it);
}

DistanceField::DistanceField(DeviceContext& ctx, int virtualWidth, int virtualHeight, float virtualDepth, int requestedSliceCount,
                             double requestedResolution, int maximumEncodedDistance, int format) : context(ctx) {
    VirtualWidth = virtualWidth; VirtualHeight = virtualHeight; VirtualDepth = virtualDepth;
    MaximumEncodedDistance = maximumEncodedDistance;
    RequestedResolution = requestedResolution;
    const Layout L = ComputeLayout(virtualWidth, virtualHeight, requestedSliceCount, requestedResolution);
    Resolution = L.Resolution;
    SliceWidth = L.SliceWidth; SliceHeight = L.SliceHeight;
    SliceCount = L.SliceCount; PhysicalSliceCount = L.PhysicalSliceCount;
    ColumnCount = L.ColumnCount; RowCount = L.RowCount;
    TextureWidth = L.TextureWidth; TextureHeight = L.TextureHeight;
    Format = format;
    ThrowIfFailed(ilm_sdf_create(ctx.Handle(), TextureWidth, TextureHeight, format, &texture));
    NeedClear = true;
    DistanceField::Invalidate();   // :120
}
DistanceField::~DistanceField() { if (texture) ilm_sdf_destroy(texture); }

// Save, :183-194
void DistanceField::Save(uint16_t* texels) const {
    if (Slices.ValidSliceCount < SliceCount)
        throw InvalidOperationException("The distance field must be fully valid");
    ThrowIfFailed(ilm_sdf_download(texture, texels));
}

// Load, :196-213
void DistanceField::Load(const uint16_t* texels) {
    ThrowIfFailed(ilm_sdf_upload(texture, texels));
    Slices.InvalidSlices.clear();
    Slices.ValidSliceCount = ((SliceCount + 2) / 3) * 3;
}

// Invalidate, :215-222
void DistanceField::Invalidate() {
    for (int i = 0; i < SliceCount; i++)
        if (!Slices.Contains(i))
            Slices.InvalidSlices.push_back(i);
}

IlmDistanceFieldRenderDesc DistanceField::GetRenderDesc(int dynamicFlagFilter) const {
    IlmDistanceFieldRenderDesc d;
    std::memset(&d, 0, sizeof(d));
    d.VirtualWidth = VirtualWidth; d.VirtualHeight = VirtualHeight; d.VirtualDepth = VirtualDepth; d.ZOffset = ZOffset;
    d.SliceWidth = SliceWidth; d.SliceHeight = SliceHeight; d.SliceCount = SliceCount;
    d.ColumnCount = ColumnCount; d.RowCount = RowCount;
    d.MaximumEncodedDistance = (float)MaximumEncodedDistance;
    d.InvScaleFactorX = (float)((double)VirtualWidth / SliceWidth);      // Uniforms.cs:108-109
    d.InvScaleFactorY = (float)((double)VirtualHeight / SliceHeight);
    d.DynamicFlagFilter = dynamicFlagFilter;
    return d;
}

// DynamicDistanceField, SDF/DistanceField.cs:248-321
DynamicDistanceField::DynamicDistanceField(DeviceContext& ctx, int virtualWidth, int virtualHeight, float virtualDepth, int sliceCount,
                                           double requestedResolution, int maximumEncodedDistance, int format)
    : DistanceField(ctx, virtualWidth, virtualHeight, virtualDepth, sliceCount, requestedResolution, maximumEncodedDistance, format) {
    ThrowIfFailed(ilm_sdf_create(ctx.Handle(), TextureWidth, TextureHeight, format, &staticTexture));
    Invalidate(true);
}
DynamicDistanceField::~DynamicDistanceField() { if (staticTexture) ilm_sdf_destroy(staticTexture); }
void DynamicDistanceField::Invalidate(bool invalidateStatic) {   // :270-277
    for (int i = 0; i < SliceCount; i++) {
        if (!Slices.Contains(i)) Slices.InvalidSlices.push_back(i);
        if (invalidateStatic && !StaticSliceInfo.Contains(i)) StaticSliceInfo.InvalidSlices.push_back(i);
    }
}
void DynamicDistanceField::ValidateSlice(int index, bool dynamic) {   // :279-285
    if (dynamic) {
        if (!StaticSliceInfo.Contains(index)) Slices.Remove(index);
    } else
        StaticSliceInfo.Remove(index);
}
void DynamicDistanceField::MarkValidSlice(int index, bool dynamic) {   // :287-292
    if (dynamic)
        Slices.ValidSliceCount = std::min(std::max(Slices.ValidSliceCount, index), StaticSliceInfo.ValidSliceCount);
    else
        StaticSliceInfo.ValidSliceCount = std::max(StaticSliceInfo.ValidSliceCount, index);
}

IlmDistanceFieldUniforms DistanceField::GetUniforms() const {
    IlmDistanceFieldUniforms u;
    std::memset(&u, 0, sizeof(u));
    u.Extent = { (float)VirtualWidth, (float)VirtualHeight, VirtualDepth, (float)MaximumEncodedDistance };   // GetExtent4
    const float sliceZSize = VirtualDepth / SliceCount;
    u.TextureSliceCount = { (float)ColumnCount, (float)RowCount, std::min(Slices.ValidSliceCount, SliceCount) * sliceZSize, (float)SliceCount };
    u.TextureSliceAndTexelSize = { 1.0f / ColumnCount, 1.0f / RowCount, 1.0f / (VirtualWidth * ColumnCount), 1.0f / (VirtualHeight * RowCount) };
    u.ConeAndMisc = { 0, 0, 0, (float)((double)VirtualWidth / SliceWidth) };
    u.StepAndMisc2 = { 0, 0, 1, (float)((double)VirtualHeight / SliceHeight) };
    return u;
}

namespace Particles {

// ---- ParticleEngine ---------------------------------------------------------------------------------------
ParticleEngine::ParticleEngine(DeviceContext& ctx, const ParticleEngineConfiguration& configuration, const float* randomnessTexels)
    : Context(ctx), Configuration(configuration) {
    ThrowIfFailed(ilm_engine_create(ctx.Handle(), configuration.ChunkSize, reinterpret_cast<const IlmFloat4*>(randomnessTexels),
                                    RandomnessTextureWidth, RandomnessTextureHeight, &handle));
}
ParticleEngine::~ParticleEngine() { if (handle) ilm_engine_destroy(handle); }

// ClampedBezier1(BezierF), Bezier.cs:442-460
IlmClampedBezier1 MakeClampedBezier1(const std::optional<BezierF>& src) {
    IlmClampedBezier1 r;
    if (!src) { r.RangeAndCount = { 0, 1, 1, 0 }; r.ABCD = { 1, 1, 1, 1 }; return r; }   // ClampedBezier1.One
    float range = src->MaxValue - src->MinValue;
    if ((range == 0) || (src->Count <= 1)) range = 1;
    r.RangeAndCount = { std::min(src->MinValue, src->MaxValue), 1.0f / range, (float)src->Count, (float)src->Mode };
    r.ABCD = { src->A, src->B, src->C, src->D };
    return r;
}
// ClampedBezier4(IBezier...), Bezier.cs:741-757
IlmClampedBezier4 MakeClampedBezier4(const std::optional<Bezier4>& src) {
    IlmClampedBezier4 r;
    const IlmFloat4 one = { 1, 1, 1, 1 };
    if (!src) { r.RangeAndCount = { 0, 1, 1, 0 }; r.A = r.B = r.C = r.D = one; return r; }   // ClampedBezier4.One
    float range = src->MaxValue - src->MinValue;
    if ((range == 0) || (src->Count <= 1)) range = 1;
    r.RangeAndCount = { std::min(src->MinValue, src->MaxValue), 1.0f / range, (float)src->Count, (float)src->Mode };
    auto cv = [](const Vector4& v) { return IlmFloat4{ v.X, v.Y, v.Z, v.W }; };
    r.A = cv(src->A); r.B = cv(src->B); r.C = cv(src->C); r.D = cv(src->D);
    return r;
}

namespace Transforms {

// ParticleAreaTransform.SetParameters, ParticleTransform.cs:294-318
void ParticleAreaTransform::FillArea(IlmAreaParams& a) const {
    std::memset(&a, 0, sizeof(a));
    if (Area) {
        a.AreaType = (int)Area->Type;
        a.AreaCenter[0] = Area->Center.X; a.AreaCenter[1] = Area->Center.Y; a.AreaCenter[2] = Area->Center.Z;
        a.AreaSize[0] = Area->Size.X; a.AreaSize[1] = Area->Size.Y; a.AreaSize[2] = Area->Size.Z;
        a.AreaFalloff = std::max(1.0f, Area->Falloff);
        a.AreaRotation = Area->Rotation;
    } else {
        a.AreaType = 0;   // AreaFalloff stays at the effect default 0
    }
    a.Strength = Strength;
    const Vector2 cf = CategoryFilter.value_or(Vector2{ -9999, 9999 });
    a.CategoryFilter[0] = cf.X; a.CategoryFilter[1] = cf.Y;
}

// FMA.SetParameters, Transforms.cs:38-45
bool FMA::FillOp(IlmTransformOp& op, double) {
    std::memset(&op, 0, sizeof(op));
    op.Type = ILM_OP_FMA;
    IlmFMAParams& p = op.u.FMA;
    FillArea(p.Area);
    p.TimeDivisor = CyclesPerSecond ? (1000 / *CyclesPerSecond) : -1.0f;
    p.PositionAdd = { Position.Add.X, Position.Add.Y, Position.Add.Z, 0 };
    p.PositionMultiply = { Position.Multiply.X, Position.Multiply.Y, Position.Multiply.Z, 1 };
    p.VelocityAdd = { Velocity.Add.X, Velocity.Add.Y, Velocity.Add.Z, 0 };
    p.VelocityMultiply = { Velocity.Multiply.X, Velocity.Multiply.Y, Velocity.Multiply.Z, 1 };
    return true;
}

// Noise, Transforms.cs:176-273
Noise::Noise(uint64_t seed) : RNG(seed) { Reset(); }
void Noise::CycleUVs() {
    CurrentU = NextU; CurrentV = NextV;
    NextU = RNG.NextDouble(); NextV = RNG.NextDouble();
}
void Noise::Reset() { LastUChangeWhen = 0; CycleUVs(); }
void Noise::AutoCycleUV(float now, double intervalSecs, float& t) {
    if (intervalSecs <= 0.01) { t = 0; return; }
    double nextChangeWhen = LastUChangeWhen + intervalSecs;
    if (now >= nextChangeWhen) {
        const double elapsed = now - nextChangeWhen;
        if (elapsed >= intervalSecs) LastUChangeWhen = now;
        else LastUChangeWhen = nextChangeWhen;
        nextChangeWhen = LastUChangeWhen + intervalSecs;
        CycleUVs();
    }
    t = (float)((now - LastUChangeWhen) / intervalSecs);
}
bool Noise::FillOp(IlmTransformOp& op, double now) {
    std::memset(&op, 0, sizeof(op));
    op.Type = ILM_OP_NOISE;
    IlmNoiseParams& p = op.u.Noise;
    FillArea(p.Area);
    p.TimeDivisor = CyclesPerSecond ? (1000 / *CyclesPerSecond) : -1.0f;
    p.PositionOffset = { Position.Offset.X, Position.Offset.Y, Position.Offset.Z, Position.Offset.W };
    p.PositionMinimum = { Position.Minimum.X, Position.Minimum.Y, Position.Minimum.Z, Position.Minimum.W };
    p.PositionScale = { Position.Scale.X, Position.Scale.Y, Position.Scale.Z, Position.Scale.W };
    p.VelocityOffset = { Velocity.Offset.X, Velocity.Offset.Y, Velocity.Offset.Z, Speed.Offset };
    p.VelocityMinimum = { Velocity.Minimum.X, Velocity.Minimum.Y, Velocity.Minimum.Z, Speed.Minimum };
    p.VelocityScale = { Velocity.Scale.X, Velocity.Scale.Y, Velocity.Scale.Z, Speed.Scale };
    const double intervalSecs = Interval / (double)IntervalUnit;
    float t;
    AutoCycleUV((float)now, intervalSecs, t);
    p.RandomnessOffset[0] = (float)(CurrentU * 253); p.RandomnessOffset[1] = (float)(CurrentV * 127);
    p.NextRandomnessOffset[0] = (float)(NextU * 253); p.NextRandomnessOffset[1] = (float)(NextV * 127);
    p.FrequencyLerp = t;
    p.ReplaceOldVelocity = ReplaceOldVelocity ? 1.0f : 0.0f;
    return true;
}

// Gravity.SetParameters, Transforms.cs:347-365
bool Gravity::FillOp(IlmTransformOp& op, double) {
    if ((int)Attractors.size() > MaxAttractors)
        throw std::runtime_error("Maximum number of attractors per instance is " + std::to_string(MaxAttractors));
    std::memset(&op, 0, sizeof(op));
    op.Type = ILM_OP_GRAVITY;
    IlmGravityParams& p = op.u.Gravity;
    p.AttractorCount = (int)Attractors.size();
    p.MaximumAcceleration = MaximumAcceleration;
    p.CategoryFilter[0] = CategoryFilter.X; p.CategoryFilter[1] = CategoryFilter.Y;
    for (size_t i = 0; i < Attractors.size(); i++) {
        const Attractor& a = Attractors[i];
        p.AttractorPositions[i][0] = a.Position.X; p.AttractorPositions[i][1] = a.Position.Y; p.AttractorPositions[i][2] = a.Position.Z;
        p.AttractorRadiusesAndStrengths[i][0] = a.Radius;
        p.AttractorRadiusesAndStrengths[i][1] = a.Strength;
        p.AttractorRadiusesAndStrengths[i][2] = (float)(int)a.Type;
    }
    return true;
}

// ---- SpawnerBase, ParticleSpawner.cs:16-260 ---------------------------------------------------------------
static IlmMatrix IdentityMatrix() {
    IlmMatrix m;
    std::memset(&m, 0, sizeof(m));
    m.m[0] = m.m[5] = m.m[10] = m.m[15] = 1;
    return m;
}
SpawnerBase::SpawnerBase(uint64_t seed) : PositionPostMatrix(IdentityMatrix()), VelocityPostMatrix(IdentityMatrix()), RNG(seed) {}

void SpawnerBase::BeginTick(double, double deltaTimeSeconds, int& spawnCount) {
    if (!IsActive || !IsActive2) {
        RateError = 0;
        spawnCount = 0;
        return;
    }
    const int countScaler = CountScale();
    float minRate = MinRate, maxRate = MaxRate;
    if (minRate > maxRate)
        minRate = maxRate;
    double currentRate = ((NextRateDraw() * (maxRate - minRate)) + minRate) * countScaler * deltaTimeSeconds;
    currentRate += RateError;
    RateError = 0;
    currentRate = AdjustCurrentRate(currentRate);
    if (currentRate < 1) {
        RateError = std::max(currentRate, 0.0);
        spawnCount = 0;
    } else {
        spawnCount = (int)currentRate;
        RateError = currentRate - spawnCount;
    }
    if (MaximumTotal) {
        const int scaledTotal = *MaximumTotal * CountScale();
        const int remaining = scaledTotal - totalSpawned;
        if (spawnCount > remaining) {
            spawnCount = remaining;
            RateError = 0;
        }
    }
}

double SpawnerBase::NextRateDraw() {
    if (!ScriptedDraws.empty()) {
        const double d = ScriptedDraws.front();
        ScriptedDraws.erase(ScriptedDraws.begin());
        return d;
    }
    return RNG.NextDouble();
}

void SpawnerBase::EndTick(int requestedSpawnCount, int actualSpawnCount) {
    RateError += requestedSpawnCount - actualSpawnCount;
    totalSpawned += actualSpawnCount;
}

float SpawnerBase::EstimateMaximumLifeForNewParticle() const {
    const float a = Life.Constant + (Life.Offset * Life.RandomScale);
    const float b = Life.Constant - (Life.Offset * Life.RandomScale);
    return std::max(a, b);
}

void SpawnerBase::FillSpawn(IlmSpawnParams& p, int chunkSize, double) {
    std::memset(&p, 0, sizeof(p));
    const double a = RNG.NextDouble(), b = RNG.NextDouble();
    p.RandomnessOffset[0] = (float)(a * 253);
    p.RandomnessOffset[1] = (float)(b * 127);
    // GetChunkSizeAndIndices, :142-146 (w filled by the subclass)
    p.ChunkSizeAndIndices[0] = (float)chunkSize;
    p.ChunkSizeAndIndices[1] = (float)indexFirst;
    p.ChunkSizeAndIndices[2] = (float)indexLast;
    p.ChunkSizeAndIndices[3] = 0;
    p.Configuration[0] = { Position.RandomScale.X, Position.RandomScale.Y, Position.RandomScale.Z, Life.RandomScale };
    p.Configuration[1] = { Position.Offset.X, Position.Offset.Y, Position.Offset.Z, Life.Offset };
    p.Configuration[2] = { Velocity.Constant.X, Velocity.Constant.Y, Velocity.Constant.Z, Category.Constant };
    p.Configuration[3] = { Velocity.RandomScale.X, Velocity.RandomScale.Y, Velocity.RandomScale.Z, Category.RandomScale };
    p.Configuration[4] = { Velocity.Offset.X, Velocity.Offset.Y, Velocity.Offset.Z, Category.Offset };
    p.Configuration[5] = { Color.Constant.X, Color.Constant.Y, Color.Constant.Z, Color.Constant.W };
    p.Configuration[6] = { Color.RandomScale.X, Color.RandomScale.Y, Color.RandomScale.Z, Color.RandomScale.W };
    p.Configuration[7] = { Color.Offset.X, Color.Offset.Y, Color.Offset.Z, Color.Offset.W };
    p.FormulaTypes[0] = (float)(int)Position.Type;
    p.FormulaTypes[1] = (float)(int)Velocity.Type;
    p.AlignVelocityAndPosition = (AlignVelocityAndPosition && Position.Circular() && Velocity.Circular()) ? 1.0f : 0.0f;
    p.AxisMask[0] = AxisMask.X; p.AxisMask[1] = AxisMask.Y; p.AxisMask[2] = AxisMask.Z;
    p.PositionMatrix = PositionPostMatrix;
    p.VelocityMatrix = VelocityPostMatrix;
    p.AttributeDiscardThreshold = AlphaDiscardThreshold / 255.0f;
    // single-position defaults; Spawner overrides
    p.PositionConstantCount = 1;
    p.InlinePositionConstants[0] = { Position.Constant.X, Position.Constant.Y, Position.Constant.Z, Life.Constant };
}

// ---- Spawner, ParticleSpawner.cs:262-419 ------------------------------------------------------------------
int Spawner::CountScale() const {
    return std::max(RatePerPosition ? (int)AdditionalPositions.size() + (PolygonLoop ? 1 : 0) : 1, 1);
}

void Spawner::FillSpawn(IlmSpawnParams& p, int chunkSize, double now) {
    SpawnerBase::FillSpawn(p, chunkSize, now);
    int count = 1 + (int)AdditionalPositions.size();
    // GetChunkSizeAndIndices, :361-374
    {
        int c = count;
        const float polygonRate = PolygonRate.value_or(0);
        if (polygonRate >= 1) {
            if (!PolygonLoop && (c > 1)) c -= 1;
            p.ChunkSizeAndIndices[3] = std::fmod(totalSpawned / polygonRate, (float)c);
        } else {
            p.ChunkSizeAndIndices[3] = (float)(totalSpawned % c);
        }
    }
    // BeginTick's Temp3, :338-346
    for (int i = 0; (i < (int)AdditionalPositions.size()) && (i < MaxInlinePositions - 1); i++) {
        const Vector3& ap = AdditionalPositions[(size_t)i];
        p.InlinePositionConstants[i + 1] = { ap.X, ap.Y, ap.Z, Life.Constant };
    }
    p.PositionConstantCount = (float)count;
    p.PolygonRate = PolygonRate.value_or(0);
    p.PolygonLoop = PolygonLoop ? 1.0f : 0.0f;
    // InitConfiguration, :393-403
    p.Configuration[8] = { VelocityAlongPolygon.Constant, VelocityAlongPolygon.RandomScale, VelocityAlongPolygon.Offset, 0 };
    p.FormulaTypes[3] = 0;
}

// GetMaterial (:295-299): SpawnFromPositionTexture once the inline constants no longer hold the positions; BeginTick's
// Temp4 (:326-345) is the PositionBuffer content: (Position.Constant, life), then (AdditionalPositions[i], life)
void Spawner::FillRecord(IlmSpawnRecord& rec, std::vector<IlmFloat4>& positions, int chunkSize, double now) {
    positions.clear();
    FillSpawn(rec.Params, chunkSize, now);
    const int count = 1 + (int)AdditionalPositions.size();
    if (count > MaxInlinePositions) {
        rec.Kind = ILM_SPAWN_POSITION_BUFFER;
        positions.push_back({ Position.Constant.X, Position.Constant.Y, Position.Constant.Z, Life.Constant });
        for (const Vector3& ap : AdditionalPositions)
            positions.push_back({ ap.X, ap.Y, ap.Z, Life.Constant });
    } else {
        rec.Kind = ILM_SPAWN_INLINE;
    }
}

// ---- FeedbackSpawner, SpecialSpawners.cs:243-443 -----------------------------------------------------------
void FeedbackSpawner::BeginTick(ParticleSystem& system, double now, double deltaTimeSeconds, int& spawnCount, int& sourceChunkIndex) {
    if (InstanceMultiplier < 1)
        InstanceMultiplier = 1;
    spawnCount = 0;
    sourceChunkIndex = -1;
    currentFeedbackSource = -1;
    if (!SourceSystem)
        return;
    // "FIXME: Support using the same system as a feedback input?" (:347-349)
    if (SourceSystem == &system)
        return;
    SpawnerBase::BeginTick(now, deltaTimeSeconds, spawnCount);

    if ((spawnCount < InstanceMultiplier) && !SpawnFromEntireWindow) {
        AddError(spawnCount);
        spawnCount = 0;
        return;
    }
    const int instances = spawnCount / InstanceMultiplier;
    const int rounded = instances * InstanceMultiplier;
    if (rounded < spawnCount) {
        if (rounded > 0) {
            AddError(spawnCount - rounded);
            spawnCount = rounded;
        }
    }
    sourceChunkIndex = SourceSystem->PickSourceForFeedback(instances);
    if (sourceChunkIndex < 0) {
        spawnCount = 0;
        return;
    }
    ParticleSystem::Chunk& sourceChunk = SourceSystem->ChunkAt(sourceChunkIndex);
    const int windowSize = SlidingWindowSize.value_or(999999);
    int availableForFeedback = sourceChunk.AvailableForFeedback();
    if (sourceChunk.NoLongerASpawnTarget) {
        const int currentWriteChunk = SourceSystem->GetCurrentSpawnTarget(false);
        if (currentWriteChunk >= 0)
            availableForFeedback += SourceSystem->ChunkAt(currentWriteChunk).AvailableForFeedback();
    }
    const int windowedAvailable = std::min(availableForFeedback, windowSize);
    const int skipAmount = std::max(0, availableForFeedback - windowedAvailable);
    sourceChunk.SkipFeedbackInput(skipAmount);
    const int availableLessMargin = std::max(0, windowedAvailable - SlidingWindowMargin);
    const int maximumPossibleSpawns = availableLessMargin * InstanceMultiplier;
    spawnCount = std::min(spawnCount, maximumPossibleSpawns);
    spawnCount = std::min(spawnCount, sourceChunk.AvailableForFeedback() * InstanceMultiplier);
    currentFeedbackSource = sourceChunkIndex;
    currentFeedbackSourceIndex = sourceChunk.FeedbackSourceIndex();
    if (SpawnFromEntireWindow) {
        const int sourceCount = std::max(spawnCount / InstanceMultiplier, 1);
        const int maxOffset = availableLessMargin - sourceCount;
        if (maxOffset > 1)
            currentFeedbackSourceIndex += (int)(RNG.NextDouble() * maxOffset);   // RNG.Next(0, maxOffset)
    }
}

// RunSpawner, ParticleSpawning.cs:159-166
void FeedbackSpawner::OnSpawned(int spawnCount) {
    if (currentFeedbackSource < 0 || !SourceSystem || SpawnFromEntireWindow)
        return;
    const int consumedCount = std::max(spawnCount / InstanceMultiplier, 1);
    SourceSystem->ChunkAt(currentFeedbackSource).TotalConsumedForFeedback += consumedCount;
}

// SetParameters, :411-427
void FeedbackSpawner::FillRecord(IlmSpawnRecord& rec, std::vector<IlmFloat4>& positions, int chunkSize, double now) {
    positions.clear();
    FillSpawn(rec.Params, chunkSize, now);
    rec.Kind = ILM_SPAWN_FEEDBACK;
    rec.Params.PositionConstantCount = 1;
    rec.Params.InlinePositionConstants[0] = { Position.Constant.X, Position.Constant.Y, Position.Constant.Z, Life.Constant };
    IlmFeedbackParams& f = rec.Feedback;
    f.SourceSystem = SourceSystem->Handle();
    f.SourceChunkIndex = currentFeedbackSource;
    f.FeedbackSourceIndex = (float)currentFeedbackSourceIndex;
    f.InstanceMultiplier = (float)InstanceMultiplier;
    f.SourceVelocityFactor = SourceVelocityFactor;
    f.AlignPositionConstant = AlignPositionConstant ? 1.0f : 0.0f;
    f.MultiplyLife = MultiplyLife ? 1.0f : 0.0f;
    f.MultiplyAttributeConstant = MultiplyColorConstant ? 1.0f : 0.0f;
    f.SourceLifeRange[0] = SourceLifeRange.X; f.SourceLifeRange[1] = SourceLifeRange.Y;
}

// ---- PatternSpawner, SpecialSpawners.cs:15-264 -----------------------------------------------------------
// Arithmetic.NextPowerOfTwo lives in Fracture (not in the tree): smallest power of two >= v; 0 for v <= 0, which
// BeginTick's `minCount <= 0` guard (:189-194) turns into "nothing to spawn"
static int NextPowerOfTwo(int v) {
    if (v <= 0) return 0;
    int r = 1;
    while (r < v) r <<= 1;
    return r;
}
void PatternSpawner::SetTexture(int width, int height, int levels, std::vector<IlmFloat4> texels) {
    size_t total = 0;
    for (int l = 0, w = width, h = height; l < levels; l++) {
        total += (size_t)w * (size_t)h;
        w = std::max(1, w >> 1); h = std::max(1, h >> 1);
    }
    if (levels < 0 || (levels > 0 && (width < 1 || height < 1 || texels.size() != total)))
        throw ArgumentException("texture levels do not match width x height");
    texWidth = width; texHeight = height; texLevels = levels;
    texData = std::move(texels);
    texVersion++;
}
Vector2 PatternSpawner::DirectTextureSize() const {
    Vector2 result{0, 0};
    if (texLevels <= 0) return result;
    result = Vector2{ (float)texWidth, (float)texHeight };
    if (TextureSizePx) {
        if (TextureSizePx->X > 0) result.X = TextureSizePx->X;
        if (TextureSizePx->Y > 0) result.Y = TextureSizePx->Y;
    }
    if (TextureTopLeftPx) { result.X -= TextureTopLeftPx->X; result.Y -= TextureTopLeftPx->Y; }
    return result;
}
int PatternSpawner::ParticlesPerRow() const { return NextPowerOfTwo((int)DirectTextureSize().X / divisor); }
int PatternSpawner::RowsPerInstance() const { return NextPowerOfTwo((int)DirectTextureSize().Y / divisor); }

// :141-160
double PatternSpawner::AdjustCurrentRate(double rate) {
    if (WholeSpawn && (totalSpawned == 0) && MaximumTotal && (*MaximumTotal > 0) && (rate >= 1) && InstantInitialSpawn) {
        const double result = std::max((double)ParticlesPerInstance(), rate);
        const double delta = rate - result;
        RateError += delta;      // bias the error down so another instance does not follow immediately
        return result;
    }
    return rate;
}

// :162-206
void PatternSpawner::BeginTick(ParticleSystem&, double now, double deltaTimeSeconds, int& spawnCount, int& sourceChunkIndex) {
    sourceChunkIndex = -1;
    if (texLevels <= 0) { spawnCount = 0; return; }
    SpawnerBase::BeginTick(now, deltaTimeSeconds, spawnCount);
    const int minCount = WholeSpawn ? ParticlesPerInstance() : ParticlesPerRow();
    if (minCount <= 0) { spawnCount = 0; return; }
    const int requestedSpawnCount = spawnCount;
    if (spawnCount < minCount) {
        AddError(spawnCount);
        spawnCount = 0;
    } else {
        spawnCount = (spawnCount / minCount) * minCount;
        AddError(requestedSpawnCount - spawnCount);
    }
}

// SetParameters, :208-256
void PatternSpawner::FillRecord(IlmSpawnRecord& rec, std::vector<IlmFloat4>& positions, int chunkSize, double now) {
    positions.clear();
    FillSpawn(rec.Params, chunkSize, now);
    rec.Kind = ILM_SPAWN_PATTERN;
    const int rowsPerInstance = RowsPerInstance();
    const int currentRow = WholeSpawn ? 0 : (rowsSpawned++) % rowsPerInstance;
    if (WholeSpawn)
        rowsSpawned = 0;
    IlmPatternParams& p = rec.Pattern;
    p.StepWidthAndSizeScale[0] = (float)divisor; p.StepWidthAndSizeScale[1] = (float)ParticlesPerRow();
    p.StepWidthAndSizeScale[2] = divisor / (float)texWidth; p.StepWidthAndSizeScale[3] = divisor / (float)texHeight;
    float baseX = 0, baseY = 0;
    if (TextureTopLeftPx) {
        baseX = TextureTopLeftPx->X / texWidth;
        baseY = TextureTopLeftPx->Y / texHeight;
    }
    p.YOffsetsAndCoordScale[0] = (float)currentRow;
    p.YOffsetsAndCoordScale[1] = (float)((currentRow * divisor) / texHeight);      // integer division in the reference (:237)
    p.YOffsetsAndCoordScale[2] = (float)divisor; p.YOffsetsAndCoordScale[3] = (float)divisor;
    p.TexelOffsetAndMipBias[0] = -0.5f / texWidth + baseX;
    p.TexelOffsetAndMipBias[1] = -0.5f / texHeight + baseY;
    p.TexelOffsetAndMipBias[2] = 0;
    p.TexelOffsetAndMipBias[3] = (float)(std::log((double)divisor) / std::log(2.0)) + MipBiasBase;     // Math.Log(Divisor, 2)
    const Vector2 size = DirectTextureSize();
    p.CenteringOffset[0] = size.X * -0.5f; p.CenteringOffset[1] = size.Y * -0.5f;
    p.MultiplyAttributeConstant = MultiplyColorConstant ? 1.0f : 0.0f;
    rec.Params.PositionConstantCount = 1;
    rec.Params.InlinePositionConstants[0] = { Position.Constant.X, Position.Constant.Y, Position.Constant.Z, Life.Constant };
}

void PatternSpawner::BindResources(ParticleSystem& system, int slot) {
    if (system.PatternBound(slot, this, texVersion)) return;
    ThrowIfFailed(ilm_system_set_spawn_pattern(system.Handle(), slot, texData.data(), texWidth, texHeight, texLevels));
}

// MatrixMultiply, Transforms.cs:52-71
MatrixMultiply::MatrixMultiply() : Position(IdentityMatrix()), Velocity(IdentityMatrix()) {}
bool MatrixMultiply::FillOp(IlmTransformOp& op, double) {
    std::memset(&op, 0, sizeof(op));
    op.Type = ILM_OP_MATRIX_MULTIPLY;
    IlmMatrixMultiplyParams& p = op.u.MatrixMultiply;
    FillArea(p.Area);
    p.TimeDivisor = CyclesPerSecond ? (1000 / *CyclesPerSecond) : -1.0f;
    p.PositionMatrix = Position;
    p.VelocityMatrix = Velocity;
    return true;
}

// SpatialNoise.SetParameters, Transforms.cs:288-293 on top of Noise's
bool SpatialNoise::FillOp(IlmTransformOp& op, double now) {
    IlmTransformOp base;
    Noise::FillOp(base, now);
    const IlmNoiseParams noise = base.u.Noise;
    std::memset(&op, 0, sizeof(op));
    op.Type = ILM_OP_SPATIAL_NOISE;
    op.u.SpatialNoise.Noise = noise;
    op.u.SpatialNoise.SpaceScale[0] = 1.0f / SpaceScale.X;
    op.u.SpatialNoise.SpaceScale[1] = 1.0f / SpaceScale.Y;
    return true;
}

}  // namespace Transforms

// ---- ParticleSystem ----------------------------------------------------------------------------------------
ParticleSystem::ParticleSystem(ParticleEngine& engine, const ParticleSystemConfiguration& configuration)
    : Engine(engine), Configuration(configuration) {
    ThrowIfFailed(ilm_system_create(engine.Handle(), &handle));
    std::memset(&lastStep, 0, sizeof(lastStep));
}
ParticleSystem::~ParticleSystem() { if (handle) ilm_system_destroy(handle); }

// CreateChunk, ParticleSystem.cs:393-415
int ParticleSystem::CreateChunk() {
    if ((int)chunks.size() >= MaxChunkCount)
        return -1;
    int32_t index = -1;
    ThrowIfFailed(ilm_system_add_chunk(handle, &index));
    Chunk c;
    c.ID = nextChunkId++;
    chunks.push_back(c);
    return (int)chunks.size() - 1;
}

// InitializeNewChunks, ParticleSpawning.cs:13-59
int ParticleSystem::Spawn(int particleCount, const IlmFloat4* positions, const IlmFloat4* velocities, const IlmFloat4* colors) {
    const int mc = ChunkMaximumCount();
    const int numToSpawn = (int)std::ceil((double)particleCount / mc);
    for (int i = 0; i < numToSpawn; i++) {
        const int ci = CreateChunk();
        if (ci < 0)
            return 0;
        const int offset = i * mc;
        const int n = std::min(mc, particleCount - offset);
        ThrowIfFailed(ilm_chunk_upload(handle, ci, ILM_PLANE_POSITION, positions + offset, 0, n));
        ThrowIfFailed(ilm_chunk_upload(handle, ci, ILM_PLANE_VELOCITY, velocities + offset, 0, n));
        if (colors)
            ThrowIfFailed(ilm_chunk_upload(handle, ci, ILM_PLANE_ATTRIBUTES, colors + offset, 0, n));
        Chunk& c = chunks[(size_t)ci];
        c.TotalSpawned = mc;
        c.Count = mc;
        ProcessLatestLivenessInfo(c);
        // ParticleSystem.Update adds new user chunks with NoLongerASpawnTarget = true (:690-697)
        c.NoLongerASpawnTarget = true;
        TotalSpawnCount += mc;
    }
    return numToSpawn * mc;
}

// ProcessLatestLivenessInfo, ParticleLiveness.cs:47-78
void ParticleSystem::ProcessLatestLivenessInfo(Chunk& c) {
    if (!c.Count)
        return;
    if (*c.Count <= 0) c.DeadFrameCount++;
    else c.DeadFrameCount = 0;
    if (c.DeadFrameCount >= DeadFrameThreshold)
        if (std::find(chunksToReap.begin(), chunksToReap.end(), c.ID) == chunksToReap.end())
            chunksToReap.push_back(c.ID);
}

// UpdateLiveCountAndReapDeadChunks, ParticleLiveness.cs:80-105 (+ the deferred readback of
// LivenessDataReadbackWorkItem / ProcessLivenessInfoData, ParticleEngine.cs:224-252)
void ParticleSystem::UpdateLiveCountAndReapDeadChunks() {
    if (livenessPending) {
        std::vector<uint32_t> counts(std::max<size_t>(livenessChunkIds.size(), 1));
        int32_t ready = 0;
        if (BlockingLivenessReadback) {
            ThrowIfFailed(ilm_system_step_counts(handle, counts.data(), (int32_t)counts.size(), 1));
            ready = 1;
        } else {
            // like the reference's deferred readback: take the counts when the GPU has produced them, never stall
            ThrowIfFailed(ilm_system_poll_counts(handle, counts.data(), (int32_t)counts.size(), 1, &ready));
        }
        if (ready) {
            for (size_t i = 0; i < livenessChunkIds.size(); i++)
                for (Chunk& c : chunks)
                    if (c.ID == livenessChunkIds[i]) {
                        c.Count = (int)(counts[i] & 0xFFFF);   // ProcessLivenessInfoData: raw & 0xFFFF
                        ProcessLatestLivenessInfo(c);
                    }
            livenessPending = false;
        }
    }
    LiveCount = 0;
    for (const Chunk& c : chunks) {
        const int chunkCount = c.Count.value_or(0);
        if (Engine.Configuration.AccurateLivenessCounts) LiveCount += chunkCount;
        else LiveCount += (chunkCount > 0) ? 1 : 0;
    }
    for (int id : chunksToReap) {
        for (size_t i = 0; i < chunks.size(); i++)
            if (chunks[i].ID == id) {
                // Reap, ParticleLiveness.cs:120-129
                ThrowIfFailed(ilm_system_remove_chunk(handle, (int32_t)i));
                chunks.erase(chunks.begin() + (long)i);
                if (currentSpawnTarget == id) currentSpawnTarget = -1;
                break;
            }
    }
    chunksToReap.clear();
}

// PickTargetForSpawn, ParticleSpawning.cs:199-231: returns the chunk table index
int ParticleSystem::PickTargetForSpawn(bool feedback, int count, bool& needClear, bool partialSpawnAllowed) {
    int& currentTarget = feedback ? currentFeedbackSpawnTarget : currentSpawnTarget;
    int index = -1;
    for (size_t i = 0; i < chunks.size(); i++)
        if (chunks[i].ID == currentTarget) index = (int)i;
    if (index >= 0) {
        Chunk& chunk = chunks[(size_t)index];
        const int free = ChunkMaximumCount() - chunk.NextSpawnOffset;
        if (free < (partialSpawnAllowed ? 16 : count)) {
            chunk.NoLongerASpawnTarget = true;
            currentTarget = -1;
            index = -1;
        }
    }
    if (index < 0) {
        index = CreateChunk();
        if (index < 0) { needClear = false; return -1; }
        chunks[(size_t)index].IsFeedbackSource = feedback;
        currentTarget = chunks[(size_t)index].ID;
        needClear = true;
    } else {
        needClear = false;
    }
    return index;
}

// GetCurrentSpawnTarget / PickSourceForFeedback, ParticleSpawning.cs:233-265
int ParticleSystem::GetCurrentSpawnTarget(bool feedback) const {
    const int id = feedback ? currentFeedbackSpawnTarget : currentSpawnTarget;
    for (size_t i = 0; i < chunks.size(); i++)
        if (chunks[i].ID == id) return (int)i;
    return -1;
}
int ParticleSystem::PickSourceForFeedback(int count) {
    for (size_t i = 0; i < chunks.size(); i++) {
        const Chunk& c = chunks[i];
        if ((c.AvailableForFeedback() >= count / 2) && !c.IsFeedbackSource) {
            currentFeedbackSource = c.ID;
            return (int)i;
        }
    }
    return -1;
}

// RunSpawner, ParticleSpawning.cs:115-197
bool ParticleSystem::RunSpawner(Transforms::SpawnerBase& spawner, double deltaTimeSeconds, double now, bool,
                                std::vector<IlmSpawnRecord>& records, std::vector<std::vector<IlmFloat4>>& recordPositions,
                                std::vector<Transforms::SpawnerBase*>& recordSpawners) {
    int spawnCount = 0, requestedSpawnCount = 0;
    if (!spawner.IsValid())
        return false;
    int sourceChunkIndex = -1;
    spawner.BeginTick(*this, now, deltaTimeSeconds, requestedSpawnCount, sourceChunkIndex);
    if (requestedSpawnCount <= 0)
        return false;
    else if (requestedSpawnCount > ChunkMaximumCount())
        spawnCount = ChunkMaximumCount();
    else
        spawnCount = requestedSpawnCount;

    bool needClear;
    const int ci = PickTargetForSpawn(spawner.IsFeedback(), spawnCount, needClear, spawner.PartialSpawnAllowed());
    if (ci < 0)
        return false;
    Chunk& chunk = chunks[(size_t)ci];
    const int free = ChunkMaximumCount() - chunk.NextSpawnOffset;
    if (spawnCount > free) {
        if (spawner.PartialSpawnAllowed()) spawnCount = free;
        else return false;
    }
    const int first = chunk.NextSpawnOffset;
    const int last = chunk.NextSpawnOffset + spawnCount - 1;
    spawner.SetIndices(first, last);
    chunk.NextSpawnOffset += spawnCount;
    TotalSpawnCount += spawnCount;
    if (sourceChunkIndex >= 0)
        spawner.OnSpawned(spawnCount);

    spawner.EndTick(requestedSpawnCount, spawnCount);
    chunk.TotalSpawned += spawnCount;
    if (spawnCount > 0) {
        chunk.DeadFrameCount = 0;
        IlmSpawnRecord rec;
        std::memset(&rec, 0, sizeof(rec));
        rec.ChunkIndex = ci;
        std::vector<IlmFloat4> positions;
        spawner.FillRecord(rec, positions, Engine.Configuration.ChunkSize, now);
        records.push_back(rec);
        recordPositions.push_back(std::move(positions));
        recordSpawners.push_back(&spawner);
    }
    chunk.ApproximateMaximumLife = std::max(chunk.ApproximateMaximumLife, spawner.EstimateMaximumLifeForNewParticle());
    return requestedSpawnCount > spawnCount;   // isPartialSpawn
}

// SetSystemUniforms (ParticleSystem.cs:547-575) + Uniforms.ParticleSystem ctor (Uniforms.cs:207-235)
// + the rotation / life ramp / distance field binds of UpdateHandler._BeforeDraw (ParticleTransform.cs:144-161)
void ParticleSystem::FillSystemUniforms(IlmStepDesc& d, double deltaTimeSeconds) const {
    const ParticleSystemConfiguration& C = Configuration;
    const int cs = Engine.Configuration.ChunkSize;
    d.System.TexelAndSize = { 1.0f / cs, 1.0f / cs, C.Size.X, C.Size.Y };
    d.System.GlobalSettings = { (float)(deltaTimeSeconds * 1000), C.Friction, C.MaximumVelocity, C.LifeDecayPerSecond };
    if (C.Collision)
        d.System.CollisionSettings = { C.Collision->EscapeVelocity, C.Collision->BounceVelocityMultiplier, C.Collision->Distance, C.Collision->LifePenalty };
    else
        d.System.CollisionSettings = { 0, 0, 0, 0 };
    d.System.AnimationRateAndRotationAndZToY = { 0, 0, C.RotationFromVelocity ? 1.0f : 0.0f, C.ZToY };

    const float o = C.Color.OpacityFromLife.value_or(0);
    if (o != 0) {
        d.Update.ColorFromLife.A = { 1, 1, 1, 0 };
        d.Update.ColorFromLife.B = { 1, 1, 1, 1 };
        d.Update.ColorFromLife.C = { 0, 0, 0, 0 };
        d.Update.ColorFromLife.D = { 0, 0, 0, 0 };
        d.Update.ColorFromLife.RangeAndCount = { 0, 1.0f / o, 2, 0 };
    } else {
        d.Update.ColorFromLife = MakeClampedBezier4(C.Color.ColorFromLife);
    }
    d.Update.ColorFromVelocity = MakeClampedBezier4(C.Color.ColorFromVelocity);
    d.Update.SizeFromLife = MakeClampedBezier1(C.SizeFromLife);
    d.Update.SizeFromVelocity = MakeClampedBezier1(C.SizeFromVelocity);
    const float deg = 3.14159265358979323846f / 180.0f;   // MathHelper.ToRadians
    d.Update.RotationFromLifeAndIndex[0] = C.RotationFromLife * deg;
    d.Update.RotationFromLifeAndIndex[1] = C.RotationFromIndex * deg;
    d.Update.LifeRampSettings = { 0, 0, 1, 1 };   // MaybeSetLifeRampParameters without a ramp, :936-939
}

void ParticleSystem::Launch(const IlmStepDesc& d) {
    lastStep = d;
    ThrowIfFailed(ilm_system_step(handle, &d));
}

// Update, ParticleSystem.cs:630-761
ParticleSystem::UpdateResult ParticleSystem::Update(int frameIndex) {
    ITimeProvider* tp = Configuration.TimeProvider ? Configuration.TimeProvider
                        : (Engine.Configuration.TimeProvider ? Engine.Configuration.TimeProvider : &defaultTime);
    const std::optional<double> lastUpdate = lastUpdateTimeSeconds;
    const double updateError = updateErrorAccumulator;
    updateErrorAccumulator = 0;
    double now = tp->Seconds();
    currentFrameIndex++;

    if (lastFrameUpdated >= frameIndex)
        throw InvalidOperationException("Cannot update twice in a single frame");

    const std::optional<int>& ups = Engine.Configuration.UpdatesPerSecond;
    const double maxDeltaTime = std::min(std::max(Engine.Configuration.MaximumUpdateDeltaTimeSeconds, (double)(1 / 200.0f)), 10.0);
    const double tickUnit = 1.0 / std::min(std::max(ups.value_or(60), 5), 200);
    double actualDeltaTimeSeconds = tickUnit;
    if (lastUpdate)
        actualDeltaTimeSeconds = std::min(now - *lastUpdate, maxDeltaTime);

    if (ups && lastUpdate) {
        actualDeltaTimeSeconds += updateError;
        int tickCount = (int)std::floor(actualDeltaTimeSeconds / tickUnit);
        if (tickCount < 0) tickCount = 0;
        const double adjustedDeltaTime = tickCount * tickUnit;
        updateErrorAccumulator = actualDeltaTimeSeconds - adjustedDeltaTime;
        actualDeltaTimeSeconds = adjustedDeltaTime;
        if ((actualDeltaTimeSeconds <= 0) && (currentFrameIndex > 1))
            return UpdateResult{ false, (float)now };
        lastUpdateTimeSeconds = now = *lastUpdate + adjustedDeltaTime;
    } else {
        lastUpdateTimeSeconds = now;
    }
    lastFrameUpdated = frameIndex;
    actualDeltaTimeSeconds = std::min(actualDeltaTimeSeconds, maxDeltaTime);
    LastDeltaTimeSeconds = actualDeltaTimeSeconds;

    UpdateLiveCountAndReapDeadChunks();

    if (isClearPending) {
        // :702-714: Erase twice (idempotent) then reap everything
        IlmStepDesc e;
        std::memset(&e, 0, sizeof(e));
        e.ChunkCount = -1;
        e.UpdateMode = ILM_UPDATE_ERASE;
        if (!chunks.empty()) Launch(e);
        while (!chunks.empty()) {
            ThrowIfFailed(ilm_system_remove_chunk(handle, (int32_t)chunks.size() - 1));
            chunks.pop_back();
        }
        isClearPending = false;
        TotalSpawnCount = 0;
        currentSpawnTarget = currentFeedbackSpawnTarget = currentFeedbackSource = -1;
        livenessPending = false;
    }

    bool computingLiveness = false;
    if (framesUntilNextLivenessCheck-- <= 0) {
        framesUntilNextLivenessCheck = LivenessCheckInterval;
        computingLiveness = true;
    }

    // spawners first (:725-741)
    std::vector<IlmSpawnRecord> records;
    std::vector<std::vector<IlmFloat4>> recordPositions;
    std::vector<Transforms::SpawnerBase*> recordSpawners;
    for (Transforms::ParticleTransform* t : Transforms) {
        if (!t->IsSpawner()) continue;
        auto* s = static_cast<Transforms::SpawnerBase*>(t);
        if (!s->IsActive || !s->IsActive2) continue;
        const bool isPartialSpawn = RunSpawner(*s, actualDeltaTimeSeconds, now, false, records, recordPositions, recordSpawners);
        if (isPartialSpawn)
            RunSpawner(*s, actualDeltaTimeSeconds, now, true, records, recordPositions, recordSpawners);
    }
    // the PositionBuffer of a position-texture record is bound to the record slot it will occupy in its launch
    auto bindPositions = [&](int slot, size_t recordIndex) {
        const std::vector<IlmFloat4>& pl = recordPositions[recordIndex];
        if (!pl.empty())
            ThrowIfFailed(ilm_system_set_spawn_positions(handle, slot, pl.data(), (int32_t)pl.size()));
        recordSpawners[recordIndex]->BindResources(*this, slot);
    };

    // UpdateChunk (:791-856) for every chunk: transforms in list order, then exactly one update technique
    std::vector<IlmTransformOp> ops;
    for (Transforms::ParticleTransform* t : Transforms) {
        const bool shouldSkip = !t->IsActive || !t->IsActive2 || t->IsSpawner() || !t->IsValid();
        if (shouldSkip) continue;
        IlmTransformOp op;
        if (t->FillOp(op, now))
            ops.push_back(op);
    }

    IlmStepDesc d;
    std::memset(&d, 0, sizeof(d));
    d.FirstChunk = 0;
    d.ChunkCount = -1;
    FillSystemUniforms(d, actualDeltaTimeSeconds);
    int finalMode = ILM_UPDATE_POSITIONS;
    if (Configuration.Collision && Configuration.Collision->Field) {
        if (!Configuration.Collision->DistanceFieldMaximumZ)
            throw InvalidOperationException("If a distance field is active, you must set DistanceFieldMaximumZ");
        finalMode = ILM_UPDATE_WITH_DISTANCE_FIELD;
        // ParticleTransform.cs:144-152: only Uniforms.DistanceField is bound; DistanceFieldPacked1 is never set on the
        // particle effect in the reference and stays zero (the field collapses to its first slice).
        d.DistanceField = Configuration.Collision->Field->GetUniforms();
        ThrowIfFailed(ilm_system_set_distance_field(handle, Configuration.Collision->Field->Texture()));
    }

    // One launch when everything fits a descriptor; otherwise earlier launches carry the surplus spawn records
    // and transforms (pass order is preserved: all spawns, then transforms in order, update last).
    size_t spawnPos = 0, opPos = 0;
    if (chunks.empty())
        return UpdateResult{ true, (float)now };
    for (;;) {
        const size_t spawnsLeft = records.size() - spawnPos, opsLeft = ops.size() - opPos;
        const bool lastLaunch = (spawnsLeft <= ILM_MAX_SPAWNS) && (opsLeft <= ILM_MAX_OPS);
        IlmStepDesc cur = d;
        if (spawnsLeft > ILM_MAX_SPAWNS) {
            // surplus spawn records go alone, before any transform runs
            cur.SpawnCount = ILM_MAX_SPAWNS;
            for (int k = 0; k < ILM_MAX_SPAWNS; k++) { cur.Spawns[k] = records[spawnPos + (size_t)k]; bindPositions(k, spawnPos + (size_t)k); }
            spawnPos += ILM_MAX_SPAWNS;
            cur.UpdateMode = ILM_UPDATE_NONE;
            Launch(cur);
            continue;
        }
        cur.SpawnCount = (int)spawnsLeft;
        for (size_t k = 0; k < spawnsLeft; k++) { cur.Spawns[k] = records[spawnPos + k]; bindPositions((int)k, spawnPos + k); }
        spawnPos = records.size();
        const size_t nOps = std::min<size_t>(opsLeft, ILM_MAX_OPS);
        cur.OpCount = (int)nOps;
        for (size_t k = 0; k < nOps; k++) cur.Ops[k] = ops[opPos + k];
        opPos += nOps;
        if (lastLaunch) {
            cur.UpdateMode = finalMode;
            if (computingLiveness) cur.Flags |= ILM_STEP_COUNT_LIVE;
            Launch(cur);
            break;
        }
        cur.UpdateMode = ILM_UPDATE_NONE;
        Launch(cur);
    }
    for (Chunk& c : chunks)
        c.ApproximateMaximumLife -= Configuration.LifeDecayPerSecond * (float)actualDeltaTimeSeconds;

    if (Configuration.AutoReadback) {     // MaybePerformReadback(timestamp), ParticleSystem.cs:625-628
        ReadbackTimestamp = (float)now;
        ReadbackResult = PerformReadback();
    }
    if (computingLiveness) {
        // ComputeLiveness (:716-720, ParticleEngine.cs:282-386): counts are produced by the update launch itself
        livenessPending = true;
        livenessChunkIds.clear();
        for (const Chunk& c : chunks) livenessChunkIds.push_back(c.ID);
    }
    return UpdateResult{ true, (float)now };
}

// the part of FillReadbackResult that runs before its loop, ParticleReadback.cs:80-115
IlmReadbackParams ParticleSystem::GetReadbackParams() const {
    IlmReadbackParams p;
    std::memset(&p, 0, sizeof(p));
    const ParticleAppearance& ap = Configuration.Appearance;
    Vector2 pSize = Configuration.Size;
    p.TextureRegion[0] = 0; p.TextureRegion[1] = 0; p.TextureRegion[2] = 1; p.TextureRegion[3] = 1;    // Bounds.Unit
    if (ap.TextureSize) {
        const Vector2 sizeF = *ap.TextureSize;
        const Vector2 sz = ap.SizePx.value_or(sizeF);
        // Bounds.FromPositionAndSize(OffsetPx / sizeF, SizePx / sizeF)
        p.TextureRegion[0] = ap.OffsetPx.X / sizeF.X; p.TextureRegion[1] = ap.OffsetPx.Y / sizeF.Y;
        p.TextureRegion[2] = p.TextureRegion[0] + sz.X / sizeF.X; p.TextureRegion[3] = p.TextureRegion[1] + sz.Y / sizeF.Y;
        if (!ap.RelativeSize)
            pSize = { Configuration.Size.X / sizeF.X, Configuration.Size.Y / sizeF.Y };
    }
    p.Size[0] = pSize.X; p.Size[1] = pSize.Y;
    p.AnimationRate[0] = ap.AnimationRate.X; p.AnimationRate[1] = ap.AnimationRate.Y;
    p.ZToY = Configuration.ZToY;
    p.ColumnFromVelocity = ap.ColumnFromVelocity ? 1 : 0;
    p.RowFromVelocity = ap.RowFromVelocity ? 1 : 0;
    p.RotationFromVelocity = Configuration.RotationFromVelocity ? 1 : 0;
    p.SortedReadback = Configuration.SortedReadback ? 1 : 0;
    return p;
}

// MaybePerformReadback, ParticleReadback.cs:21-71
ParticleSystem::ReadbackView ParticleSystem::PerformReadbackView() const {
    ReadbackView view;
    if (chunks.empty())
        return view;
    const int cs = Engine.Configuration.ChunkSize;
    std::vector<int32_t> elements;
    for (const Chunk& c : chunks) {
        const int rowCount = (int)std::ceil(c.TotalSpawned / (float)cs);   // :57
        elements.push_back(std::min(rowCount * cs, ChunkMaximumCount()));
    }
    // ReadbackResultBuffer (:40-41) is the context's pinned buffer: sized for every examined slot, reused, never cleared
    const IlmReadbackParams p = GetReadbackParams();
    int32_t total = 0;
    ThrowIfFailed(ilm_system_readback_view(handle, elements.data(), (int32_t)elements.size(), &p, &view.Records, &total));
    view.Count = total;
    return view;
}

std::vector<IlmReadbackDrawCall> ParticleSystem::PerformReadback() const {
    const ReadbackView v = PerformReadbackView();
    return std::vector<IlmReadbackDrawCall>(v.Records, v.Records + v.Count);
}

IlmRasterizeParams ParticleSystem::GetRasterizeParams(int blendMode, const RenderParameters* rp, Vector2 viewportScale, Vector2 viewportPosition) const {
    IlmRasterizeParams p;
    std::memset(&p, 0, sizeof(p));
    const ParticleSystemConfiguration& C = Configuration;
    const Vector2 origin = rp ? rp->Origin : Vector2{0, 0}, scale = rp ? rp->Scale : Vector2{1, 1};
    const ParticleAppearance& ap = C.Appearance;
    if (ap.TextureSize) {
        // Uniforms.RasterizeParticleSystem ctor, Uniforms.cs:252-277
        const Vector2 texSize = *ap.TextureSize;
        const Vector2 sizePx = ap.SizePx.value_or(texSize);
        const Vector2 offset{ ap.OffsetPx.X / texSize.X, ap.OffsetPx.Y / texSize.Y };
        const Vector2 size{ sizePx.X / texSize.X, sizePx.Y / texSize.Y };
        p.BitmapTextureRegion = { offset.X, offset.Y, offset.X + size.X, offset.Y + size.Y };
        if (ap.RelativeSize)
            p.SizeFactorAndPosition = { sizePx.X * 0.5f, sizePx.Y * 0.5f, origin.X, origin.Y };
        else
            p.SizeFactorAndPosition = { 1, 1, origin.X, origin.Y };
        p.BitmapFilter = ap.Bilinear ? ILM_BITMAP_LINEAR : ILM_BITMAP_POINT;      // material choice, ParticleSystem.cs:963-971
    } else {
        p.BitmapTextureRegion = { 0, 0, 1, 1 };
        p.SizeFactorAndPosition = { 1, 1, origin.X, origin.Y };
        p.BitmapFilter = ILM_BITMAP_NONE;
    }
    // System.AnimationRateAndRotationAndZToY.xy, Uniforms.cs:230-234
    p.AnimationRate[0] = (ap.AnimationRate.X != 0) ? 1.0f / ap.AnimationRate.X : 0.0f;
    p.AnimationRate[1] = (ap.AnimationRate.Y != 0) ? 1.0f / ap.AnimationRate.Y : 0.0f;
    p.Scale = { scale.X, scale.Y, 0, 0 };
    Vector4 g = C.Color.Global;
    g.X *= g.W; g.Y *= g.W; g.Z *= g.W;
    p.GlobalColor = { g.X, g.Y, g.Z, g.W };
    p.ZFormula = { C.ZFormula.X, C.ZFormula.Y, C.ZFormula.Z, C.ZFormula.W };
    p.ZConfiguration = { C.SizeFromZ, 0, 0, 0 };
    p.RoundingPowerFromLife = MakeClampedBezier1(C.Appearance.RoundingPowerFromLife);
    p.RenderingOptions[0] = C.Appearance.Rounded ? 1.0f : 0.0f;
    p.RenderingOptions[1] = C.Appearance.DitheredOpacity ? 1.0f : 0.0f;
    p.RenderingOptions[2] = C.Appearance.ColumnFromVelocity ? 1.0f : 0.0f;
    p.RenderingOptions[3] = C.Appearance.RowFromVelocity ? 1.0f : 0.0f;
    p.SystemSize[0] = C.Size.X; p.SystemSize[1] = C.Size.Y;
    p.ZToY = C.ZToY;
    p.StippleFactor = (rp && rp->StippleFactor) ? *rp->StippleFactor : C.StippleFactor;
    p.ViewportScale[0] = viewportScale.X; p.ViewportScale[1] = viewportScale.Y;
    p.ViewportPosition[0] = viewportPosition.X; p.ViewportPosition[1] = viewportPosition.Y;
    p.BlendMode = blendMode;
    return p;
}

void ParticleSystem::SetBitmap(int width, int height, const IlmFloat4* texels) {
    ThrowIfFailed(ilm_system_set_bitmap(handle, texels, width, height));
}

ParticleSystem::RenderStats ParticleSystem::Render(RenderTarget& target, int blendMode, const RenderParameters* rp, Vector2 viewportScale,
                                                   Vector2 viewportPosition, bool wantStats) const {
    RenderStats st;
    if (chunks.empty())
        return st;
    std::vector<int32_t> quads;
    for (const Chunk& c : chunks)
        quads.push_back(std::min(ChunkMaximumCount(), c.TotalSpawned + 1));      // RenderChunk, :880
    const IlmRasterizeParams p = GetRasterizeParams(blendMode, rp, viewportScale, viewportPosition);
    uint64_t stats[3] = { 0, 0, 0 };
    ThrowIfFailed(ilm_render_particles(handle, quads.data(), (int32_t)quads.size(), &p, target.Handle(), wantStats ? stats : nullptr));
    st.LiveQuads = stats[0]; st.TilePairs = stats[1]; st.ShadedPixels = stats[2];
    return st;
}

void ParticleSystem::Readback(int chunkIndex, int plane, IlmFloat4* dst) const {
    ThrowIfFailed(ilm_chunk_download(handle, chunkIndex, plane, dst, 0, ChunkMaximumCount()));
}

}  // namespace Particles

namespace Lighting {

LightingRenderer::LightingRenderer(DeviceContext& ctx, const RendererConfiguration& configuration, LightingEnvironment* environment,
                                   void* externalLightmap)
    : Context(ctx), Configuration(configuration), Environment(environment), Probes(configuration.MaximumLightProbeCount) {
    // lightmap format: HalfVector4 when HighQuality else Color (LightingRenderer.cs:476-479)
    lightmapFormat = configuration.FloatLightmap ? ILM_LIGHTMAP_FLOAT4 : (configuration.HighQuality ? ILM_LIGHTMAP_HALF4 : ILM_LIGHTMAP_RGBA8);
    ThrowIfFailed(ilm_lightmap_create(ctx.Handle(), configuration.RenderWidth, configuration.RenderHeight, lightmapFormat, externalLightmap, &lightmap));
}
LightingRenderer::~LightingRenderer() {
    if (lightmap) ilm_lightmap_destroy(lightmap);
    if (gbuffer) ilm_gbuffer_destroy(gbuffer);
}

void LightingRenderer::SetGBuffer(const void* texels, int width, int height, int format) {
    if (gbuffer) { ilm_gbuffer_destroy(gbuffer); gbuffer = 0; }
    if (!texels) return;
    ThrowIfFailed(ilm_gbuffer_create(Context.Handle(), width, height, format, &gbuffer));
    ThrowIfFailed(ilm_gbuffer_upload(gbuffer, texels));
    gbufferWidth = width; gbufferHeight = height;
}

// RenderSphereLightSource, LightingRenderer.cs:1193-1219
bool LightingRenderer::PackSphereLight(const SphereLightSource& l, float intensityScale, bool haveDistanceField, IlmLightVertex& v) {
    if (l.Opacity <= 0.0f)
        return false;
    const IlmFloat4 pos = { l.Position.X, l.Position.Y, l.Position.Z, 0 };
    v.LightPosition1 = v.LightPosition2 = v.LightPosition3 = pos;
    v.Color1 = { l.Color.X, l.Color.Y, l.Color.Z, l.Color.W * (l.Opacity * intensityScale) };
    v.Color2 = { l.SpecularColor.X, l.SpecularColor.Y, l.SpecularColor.Z, l.SpecularPower };
    v.LightProperties = { l.Radius, l.RampLength, (float)(int)l.RampMode, (l.CastsShadows && haveDistanceField) ? 1.0f : 0.0f };
    v.MoreLightProperties = { l.AmbientOcclusionRadius, l.ShadowDistanceFalloff.value_or(-99999.0f), l.FalloffYFactor, l.AmbientOcclusionOpacity };
    // RampOffsetForGPU / RampRateForGPU, LightSource.cs:97-98
    v.EvenMoreLightProperties = { (float)l.ShadowFilter, 0, (float)-3.14159265358979323846 + l.RampOffset,
                                  (float)(1.0 / (3.14159265358979323846 * 2) * l.RampRate) };
    return true;
}

// SetDistanceFieldParameters, LightingRenderer.cs:1894-1940
IlmDistanceFieldUniforms LightingRenderer::GetDistanceFieldUniforms(const RendererQualitySettings& q) const {
    IlmDistanceFieldUniforms dfu;
    if (!Field) {
        std::memset(&dfu, 0, sizeof(dfu));
        dfu.ConeAndMisc.w = 1; dfu.StepAndMisc2.w = 1;      // InvScaleFactorX/Y = 1
        dfu.Extent.z = Environment->MaximumZ;
        dfu.StepAndMisc2.x = (float)q.MaxStepCount;
        dfu.StepAndMisc2.y = q.MinStepSize;
        return dfu;                                          // DistanceFieldPacked1 = 0
    }
    dfu = Field->GetUniforms();
    dfu.ConeAndMisc.x = q.MaxConeRadius;
    dfu.ConeAndMisc.y = Field->ZOffset;
    dfu.ConeAndMisc.z = q.OcclusionToOpacityPower;
    dfu.StepAndMisc2.x = (float)q.MaxStepCount;
    dfu.StepAndMisc2.y = q.MinStepSize;
    dfu.StepAndMisc2.z = q.LongStepFactor;
    dfu.Packed1 = { (float)((1.0f / std::max(0.0001f, dfu.TextureSliceCount.x)) * (1.0f / 3.0f)),
                    (float)((1.0f / std::max(0.0001f, dfu.Extent.z)) * dfu.TextureSliceCount.w),
                    dfu.TextureSliceCount.z, dfu.StepAndMisc2.y };
    return dfu;
}

IlmEnvironment LightingRenderer::GetEnvironmentUniforms() const {
    IlmEnvironment e;
    std::memset(&e, 0, sizeof(e));
    const float zToY = Configuration.TwoPointFiveD ? Environment->ZToYMultiplier : 0.0f;
    e.ZAndScale = { Environment->GroundZ, Environment->MaximumZ, Configuration.RenderScale.X, Configuration.RenderScale.Y };
    e.ZToY = { zToY, (std::fabs(zToY) <= 0.0001f) ? 0.0f : 1.0f / zToY, Configuration.LightOcclusion, 0 };   // Uniforms.cs:45-59
    if (gbuffer)
        e.GBufferTexelSizeAndMisc = { 1.0f / gbufferWidth, 1.0f / gbufferHeight, 1, 1 };
    else
        e.GBufferTexelSizeAndMisc = { 0, 0, 1, 1 };
    return e;
}

// RenderLighting, LightingRenderer.cs:917-1191 (sphere lights only)
void LightingRenderer::RenderLighting(float intensityScale, int rowBegin, int rowEnd, IlmRenderStats* stats) {
    if (rowEnd < 0) rowEnd = Configuration.RenderHeight;
    vertices.clear();
    // LightSorter (:2066-2096): SortKey first; blend mode, ramp texture and type are the same for every light of this pass.  The
    // reference's sort is not stable for equal keys; a stable one keeps list order, which is one of its possible outcomes
    // (Lights before Replicators here, since the mirror holds them in two lists).
    struct Entry { int sortKey; const SphereLightSource* sphere; const LightSourceReplicator* replicator; };
    std::vector<Entry> sorted;
    for (const SphereLightSource& l : Environment->Lights)
        if (l.Enabled) sorted.push_back({ l.SortKey, &l, nullptr });
    for (const LightSourceReplicator& r : Environment->Replicators)
        if (r.Enabled) sorted.push_back({ r.SortKey, nullptr, &r });
    std::stable_sort(sorted.begin(), sorted.end(), [](const Entry& x, const Entry& y) { return x.sortKey < y.sortKey; });
    // GetLightRenderState (:799-845): lights that share a ramp texture and quality settings form one render state (BlendState is
    // additive for every light here); the states are drawn one after the other onto the same target, in the order their keys first appear.
    // A replicator takes its state from its Template (:802).
    groupKeys.clear();
    groupQuality.clear();
    groups.clear();
    auto groupFor = [&](const SphereLightSource& l) -> size_t {
        const RampTexture* ramp = l.TextureRef ? l.TextureRef.get() : Configuration.DefaultRampTexture.get();
        if (ramp && ((ramp->Width == 1 && ramp->Height == 1) || ramp->Width <= 0))
            ramp = nullptr;                                   // a 1 x 1 ramp is no ramp (:822-827)
        const RendererQualitySettings* quality = l.Quality.get();
        size_t g = 0;
        while (g < groupKeys.size() && !(groupKeys[g] == ramp && groupQuality[g] == quality)) g++;
        if (g == groupKeys.size()) { groupKeys.push_back(ramp); groupQuality.push_back(quality); groups.emplace_back(); }
        return g;
    };
    for (const Entry& e : sorted) {
        IlmLightVertex v;
        if (e.sphere) {
            if (!PackSphereLight(*e.sphere, intensityScale, Field != nullptr, v))
                continue;
            groups[groupFor(*e.sphere)].push_back(v);
            vertices.push_back(v);
            continue;
        }
        // RenderReplicatorLightSource (:1221-1255): the template's packed vertex with position / colour / radius / ramp length /
        // specular replaced per placement; a placement whose final alpha is <= 0 is dropped
        const SphereLightSource& t = e.replicator->Template;
        SphereLightSource visible = t;
        visible.Opacity = 1;                                  // the template's own Opacity only matters through the per-placement product
        PackSphereLight(visible, intensityScale, Field != nullptr, v);
        const size_t g = groupFor(t);
        for (const ReplicatedLight& rl : e.replicator->Lights) {
            const Vector4 color = rl.Color.value_or(t.Color);
            const float alpha = color.W * (rl.Opacity.value_or(t.Opacity) * intensityScale);
            if (alpha <= 0)
                continue;
            const Vector3 spec = rl.SpecularColor.value_or(t.SpecularColor);
            const IlmFloat4 pos = { rl.Position.X, rl.Position.Y, rl.Position.Z, 0 };
            v.LightPosition1 = v.LightPosition2 = v.LightPosition3 = pos;
            v.Color1 = { color.X, color.Y, color.Z, alpha };
            v.Color2 = { spec.X, spec.Y, spec.Z, rl.SpecularPower.value_or(t.SpecularPower) };
            v.LightProperties.x = rl.Radius.value_or(t.Radius);
            v.LightProperties.y = rl.RampLength.value_or(t.RampLength);
            groups[g].push_back(v);
            vertices.push_back(v);
        }
    }
    const IlmEnvironment env = GetEnvironmentUniforms();
    const IlmDistanceFieldUniforms dfu = GetDistanceFieldUniforms(Configuration.DefaultQuality);
    // clear colour: Ambient * intensityScale (:1013-1024)
    const float ambient[4] = { Environment->Ambient.X * intensityScale, Environment->Ambient.Y * intensityScale,
                               Environment->Ambient.Z * intensityScale, Environment->Ambient.W * intensityScale };
    if (groups.empty()) { groupKeys.push_back(nullptr); groupQuality.push_back(nullptr); groups.emplace_back(); }     // no lights: the clear still happens
    if (stats) { stats->SdfSamples = stats->PixelLightPairs = stats->TracedPairs = 0; }
    for (size_t g = 0; g < groups.size(); g++) {
        BindRamp(groupKeys[g]);
        IlmRenderStats gs{};
        const IlmDistanceFieldUniforms gdfu = groupQuality[g] ? GetDistanceFieldUniforms(*groupQuality[g]) : dfu;     // SetDistanceFieldParameters(material, true, ltrs.Key.Quality), :792
        ThrowIfFailed(ilm_render_sphere_lights(Context.Handle(), groups[g].empty() ? nullptr : groups[g].data(), (int32_t)groups[g].size(),
                                               &env, &gdfu, gbuffer, Field ? Field->Texture() : 0, (g == 0) ? ambient : nullptr, lightmap, rowBegin, rowEnd,
                                               stats ? &gs : nullptr));
        if (stats) { stats->SdfSamples += gs.SdfSamples; stats->PixelLightPairs += gs.PixelLightPairs; stats->TracedPairs += gs.TracedPairs; }
    }
    BindRamp(nullptr);         // particle lights have no ramp technique (:176-178)
    // particle light sources: one more light-type render state each, blended on top (:1126-1141)
    for (const ParticleLightSource& pls : Environment->ParticleLights) {
        if (!pls.Enabled || !pls.IsActive || !pls.System)
            continue;
        const IlmParticleLightParams p = PackParticleLight(pls, Field != nullptr);
        // RenderChunk, ParticleSystem.cs:880: quadCount = min(ChunkMaximumCount, chunk.TotalSpawned + 1)
        std::vector<int32_t> quads;
        for (const Particles::ParticleSystem::Chunk& c : pls.System->Chunks())
            quads.push_back(std::min(pls.System->ChunkMaximumCount(), c.TotalSpawned + 1));
        if (quads.empty())
            continue;
        IlmRenderStats ps{};
        ThrowIfFailed(ilm_render_particle_lights(Context.Handle(), pls.System->Handle(), quads.data(), (int32_t)quads.size(), &p, &env, &dfu,
                                                 gbuffer, Field ? Field->Texture() : 0, lightmap, rowBegin, rowEnd, stats ? &ps : nullptr));
        if (stats) { stats->SdfSamples += ps.SdfSamples; stats->PixelLightPairs += ps.PixelLightPairs; stats->TracedPairs += ps.TracedPairs; }
    }
    if (Probes.Count() > 0)      // :1176-1182
        UpdateLightProbes(intensityScale);
}

// _LightBatchSetup binds the group's RampTexture (:764-766); the native layer keeps one per context
void LightingRenderer::BindRamp(const RampTexture* ramp) {
    if (ramp == boundRamp) return;
    ThrowIfFailed(ilm_ctx_set_light_ramp(Context.Handle(), ramp ? ramp->Texels.data() : nullptr, ramp ? ramp->Width : 0, ramp ? ramp->Height : 0));
    boundRamp = ramp;
}

IlmParticleLightParams LightingRenderer::PackParticleLight(const ParticleLightSource& pls, bool haveDistanceField) {
    const SphereLightSource& l = pls.Template;
    IlmParticleLightParams p;
    std::memset(&p, 0, sizeof(p));
    p.LightProperties = { l.Radius, l.RampLength, (float)(int)l.RampMode, (l.CastsShadows && haveDistanceField) ? 1.0f : 0.0f };
    p.MoreLightProperties = { l.AmbientOcclusionOpacity > 0.001 ? l.AmbientOcclusionRadius : 0.0f, l.ShadowDistanceFalloff.value_or(-99999.0f),
                              l.FalloffYFactor, std::min(std::max(l.AmbientOcclusionOpacity, 0.0f), 1.0f) };
    p.LightColor = { l.Color.X, l.Color.Y, l.Color.Z, l.Color.W };
    p.LightSpecularColor = { l.SpecularColor.X, l.SpecularColor.Y, l.SpecularColor.Z, l.SpecularPower };
    p.StippleFactor = pls.StippleFactor.value_or(1.0f);
    return p;
}

// float -> IEEE half (round to nearest even) -> float, for finite values inside the half range
static float HalfRound(float f) {
    if (!(std::fabs(f) < 65504.0f)) return f;
    if (std::fabs(f) < 6.103515625e-05f)            // below 2^-14: subnormal halves, spacing 2^-24
        return std::nearbyint(f * 16777216.0f) / 16777216.0f;
    int e;
    const float m = std::frexp(f, &e);              // f = m * 2^e, 0.5 <= |m| < 1: 11 significant bits
    return std::ldexp(std::nearbyint(m * 2048.0f) / 2048.0f, e);
}

// UpdateLightProbeTexture + UpdateLightProbes + LightProbeDownloadTask, LightingRenderer.LightProbes.cs:49-150.  The reference reads
// the values back a frame later on a worker thread; here the call synchronises and the probes hold this frame's values.
void LightingRenderer::UpdateLightProbes(float intensityScale) {
    const int n = Probes.Count();
    std::vector<IlmFloat4> positions((size_t)n), normals((size_t)n), values((size_t)n);
    for (int i = 0; i < n; i++) {
        const LightProbe& p = *Probes.Items[(size_t)i];
        positions[(size_t)i] = { p.Position.X, p.Position.Y, p.Position.Z, 1.0f };
        if (p.Normal) normals[(size_t)i] = { p.Normal->X, p.Normal->Y, p.Normal->Z, p.EnableShadows ? 1.0f : 0.0f };
        else normals[(size_t)i] = { 0, 0, 0, p.EnableShadows ? 1.0f : 0.0f };
    }
    Probes.IsDirty = false;
    const IlmEnvironment env = GetEnvironmentUniforms();
    const IlmDistanceFieldUniforms dfu = GetDistanceFieldUniforms(Configuration.DefaultQuality);
    // the light vertices of this frame were packed with intensityScale folded into Color1.a (RenderSphereLightSource, :1203); every
    // render state draws its lights onto the probe target with its own probe material (with or without a ramp, :159-164): summed here
    for (size_t g = 0; g < groups.size(); g++) {
        if (groups[g].empty() && g > 0) continue;
        BindRamp(groupKeys[g]);
        std::vector<IlmFloat4> part((size_t)n);
        const IlmDistanceFieldUniforms gdfu = groupQuality[g] ? GetDistanceFieldUniforms(*groupQuality[g]) : dfu;
        ThrowIfFailed(ilm_render_light_probes(Context.Handle(), groups[g].empty() ? nullptr : groups[g].data(), (int32_t)groups[g].size(),
                                              positions.data(), normals.data(), n, &env, &gdfu, Field ? Field->Texture() : 0, part.data()));
        for (int i = 0; i < n; i++) {
            values[(size_t)i].x += part[(size_t)i].x; values[(size_t)i].y += part[(size_t)i].y;
            values[(size_t)i].z += part[(size_t)i].z; values[(size_t)i].w += part[(size_t)i].w;
        }
    }
    BindRamp(nullptr);
    const float scaleFactor = 1.0f / intensityScale;     // LightProbeDownloadTask.ScaleFactor, LightingRenderer.cs:943
    for (int i = 0; i < n; i++) {
        LightProbe& p = *Probes.Items[(size_t)i];
        p.PreviousValue = p.Value;
        // the probe target is a HalfVector4 (LightProbes.cs:20,124-131): the read-back sees fp16 values (rounded once here; the reference's
        // ROP rounds after every light)
        p.Value = { HalfRound(values[(size_t)i].x) * scaleFactor, HalfRound(values[(size_t)i].y) * scaleFactor,
                    HalfRound(values[(size_t)i].z) * scaleFactor, HalfRound(values[(size_t)i].w) * scaleFactor };
    }
}

// ---- LightObstruction.cs ---------------------------------------------------------------------------------
LightObstruction::LightObstruction(LightObstructionType t, Vector3 c, Vector3 radius, float rotation) : type(t), center(c), size(radius) {
    SetRotation(rotation);
}
void LightObstruction::SetRotation(float value) {
    shadowRotation = value;
    // Quaternion.CreateFromAxisAngle(Vector3.UnitZ, angle): half = angle * 0.5f; (axis * sin(half), cos(half))
    const float half = value * 0.5f;
    SetOrientation({ 0.0f, 0.0f, (float)std::sin((double)half), (float)std::cos((double)half) });
}
IlmObstruction LightObstruction::Vertex() const {
    IlmObstruction o;
    o.Center[0] = center.X; o.Center[1] = center.Y; o.Center[2] = center.Z; o.Type = (int32_t)type;
    o.Size[0] = size.X; o.Size[1] = size.Y; o.Size[2] = size.Z; o.IsDynamic = isDynamic ? 1 : 0;
    o.Orientation[0] = orientation.X; o.Orientation[1] = orientation.Y; o.Orientation[2] = orientation.Z; o.Orientation[3] = orientation.W;
    return o;
}

// AutoInvalidateDistanceField, LightingRenderer.cs:1977-2014
void LightingRenderer::AutoInvalidateDistanceField() {
    DynamicDistanceField* ddf = dynamic_cast<DynamicDistanceField*>(Field);
    bool hasInvalidatedStatic = false, hasInvalidatedDynamic = false;
    LightObstructionCollection& obstructions = Environment->Obstructions;
    if (obstructions.IsInvalidDynamic) {
        if (ddf) ddf->Invalidate(false);
        hasInvalidatedDynamic = true;
    }
    if (obstructions.IsInvalid) {
        Field->Invalidate();
        hasInvalidatedStatic = true;
    }
    obstructions.IsInvalid = obstructions.IsInvalidDynamic = false;
    for (auto& obs : obstructions.Items) {
        if (obs->HasDynamicityChanged) {
            obs->HasDynamicityChanged = false;
            if (!hasInvalidatedStatic) {
                hasInvalidatedStatic = hasInvalidatedDynamic = true;
                Field->Invalidate();
            }
        }
        if (!obs->IsValid) {
            obs->IsValid = true;
            if (ddf && obs->IsDynamic()) {
                if (!hasInvalidatedDynamic) {
                    hasInvalidatedDynamic = true;
                    ddf->Invalidate(false);
                }
            } else if (!hasInvalidatedStatic) {
                hasInvalidatedStatic = hasInvalidatedDynamic = true;
                Field->Invalidate();
            }
        }
    }
}

// RenderDistanceFieldPartition, LightingRenderer.DistanceField.cs:415-464.  The reference issues one render-target pass per
// slice triplet; here the triplets of one partition go to the device as one ilm_sdf_render_slices call.
int LightingRenderer::RenderDistanceFieldPartition(int dynamicFlagFilter) {
    DynamicDistanceField* ddf = dynamic_cast<DynamicDistanceField*>(Field);
    if (!ddf) dynamicFlagFilter = -1;
    const bool isRenderingStatic = ddf && (dynamicFlagFilter == 0);
    SliceInfo& sliceInfo = isRenderingStatic ? ddf->StaticSliceInfo : Field->Slices;
    const IlmHandle renderTarget = isRenderingStatic ? ddf->StaticTexture() : Field->Texture();

    int slicesToUpdate = std::min(Configuration.MaximumFieldUpdatesPerFrame, (int)sliceInfo.InvalidSlices.size());
    if (slicesToUpdate <= 0)
        return 0;

    std::vector<int32_t> firstSlices;
    while (slicesToUpdate > 0) {
        if (sliceInfo.InvalidSlices.empty())
            break;   // (the reference would index an empty list here when the count is not a multiple of 3)
        const int slice = sliceInfo.InvalidSlices[0];
        firstSlices.push_back(slice - (slice % DistanceField::PackedSliceCount));
        // RenderDistanceFieldSliceTriplet, :137-148
        const int first = firstSlices.back(), last = first + 2;
        for (int i = first; i <= last; i++) {
            if (ddf) ddf->ValidateSlice(i, dynamicFlagFilter == 1);
            else Field->ValidateSlice(i);
        }
        if (ddf) ddf->MarkValidSlice(last + 1, dynamicFlagFilter == 1);
        else Field->MarkValidSlice(last + 1);
        slicesToUpdate -= 3;
    }

    std::vector<IlmObstruction> obstructions;
    obstructions.reserve(Environment->Obstructions.Items.size());
    for (const auto& o : Environment->Obstructions.Items)
        obstructions.push_back(o->Vertex());
    std::vector<IlmHeightVolume> volumes;
    std::vector<float> polygon;
    for (const HeightVolume& hv : Environment->HeightVolumes) {
        IlmHeightVolume v;
        std::memset(&v, 0, sizeof(v));
        v.FirstVertex = (int32_t)(polygon.size() / 2); v.VertexCount = (int32_t)hv.Polygon.size();
        v.ZBase = hv.ZBase; v.Height = hv.Height; v.IsDynamic = hv.IsDynamic ? 1 : 0; v.TopFaceEnableShadows = hv.TopFaceEnableShadows ? 1 : 0;
        for (const Vector2& p : hv.Polygon) { polygon.push_back(p.X); polygon.push_back(p.Y); }
        volumes.push_back(v);
    }
    const IlmDistanceFieldRenderDesc desc = Field->GetRenderDesc(dynamicFlagFilter);
    // ClearDistanceFieldSlice: the dynamic partition starts from the static texture (:117-119)
    const IlmHandle clearSource = (ddf && dynamicFlagFilter == 1) ? ddf->StaticTexture() : 0;
    Field->NeedClear = false;
    ThrowIfFailed(ilm_sdf_render_slices(renderTarget, clearSource, &desc, firstSlices.data(), (int32_t)firstSlices.size(),
                                        obstructions.empty() ? nullptr : obstructions.data(), (int32_t)obstructions.size(),
                                        volumes.empty() ? nullptr : volumes.data(), (int32_t)volumes.size(),
                                        polygon.empty() ? nullptr : polygon.data(), (int32_t)(polygon.size() / 2)));
    return (int)firstSlices.size();
}

// RenderGBuffer, LightingRenderer.GBuffer.cs:127-219
void LightingRenderer::RenderGBuffer(Vector2 viewportPosition, Vector2 viewportScale) {
    if (Configuration.TwoPointFiveD)
        throw InvalidOperationException("the 2.5D G-buffer (front faces, billboards) is not built");
    const int format = Configuration.HighQualityGBuffer ? ILM_GBUFFER_FLOAT4 : ILM_GBUFFER_HALF4;
    if (!gbuffer || gbufferWidth != Configuration.RenderWidth || gbufferHeight != Configuration.RenderHeight) {   // EnsureGBuffer
        if (gbuffer) { ilm_gbuffer_destroy(gbuffer); gbuffer = 0; }
        ThrowIfFailed(ilm_gbuffer_create(Context.Handle(), Configuration.RenderWidth, Configuration.RenderHeight, format, &gbuffer));
        gbufferWidth = Configuration.RenderWidth; gbufferHeight = Configuration.RenderHeight;
    }
    IlmGBufferRenderDesc d;
    std::memset(&d, 0, sizeof(d));
    d.ViewportPosition[0] = viewportPosition.X; d.ViewportPosition[1] = viewportPosition.Y;
    d.ViewportScale[0] = viewportScale.X * Configuration.RenderScale.X;     // actualScaleFactor, :131
    d.ViewportScale[1] = viewportScale.Y * Configuration.RenderScale.Y;
    d.GroundZ = Environment->GroundZ;
    d.RenderGroundPlane = Configuration.RenderGroundPlane ? 1 : 0;
    d.EnableGroundShadows = Environment->EnableGroundShadows ? 1 : 0;
    std::vector<IlmHeightVolume> volumes;
    std::vector<float> polygon;
    for (const HeightVolume& hv : Environment->HeightVolumes) {
        IlmHeightVolume v;
        std::memset(&v, 0, sizeof(v));
        v.FirstVertex = (int32_t)(polygon.size() / 2); v.VertexCount = (int32_t)hv.Polygon.size();
        v.ZBase = hv.ZBase; v.Height = hv.Height; v.IsDynamic = hv.IsDynamic ? 1 : 0; v.TopFaceEnableShadows = hv.TopFaceEnableShadows ? 1 : 0;
        for (const Vector2& p : hv.Polygon) { polygon.push_back(p.X); polygon.push_back(p.Y); }
        volumes.push_back(v);
    }
    ThrowIfFailed(ilm_gbuffer_render(gbuffer, &d, volumes.empty() ? nullptr : volumes.data(), (int32_t)volumes.size(),
                                     polygon.empty() ? nullptr : polygon.data(), (int32_t)(polygon.size() / 2)));
}

// UpdateFields, LightingRenderer.cs:1949-1975 + RenderDistanceField, LightingRenderer.DistanceField.cs:20-30
int LightingRenderer::UpdateFields() {
    if (Configuration.EnableGBuffer)
        RenderGBuffer();
    if (!Field)
        return 0;
    AutoInvalidateDistanceField();
    if (!Field->NeedsRasterize())
        return 0;
    int rendered = 0;
    if (dynamic_cast<DynamicDistanceField*>(Field)) {
        rendered += RenderDistanceFieldPartition(0);
        rendered += RenderDistanceFieldPartition(1);
    } else
        rendered += RenderDistanceFieldPartition(-1);
    return rendered;
}

void LightingRenderer::Resolve(IlmHandle destination, const IlmHDRConfiguration* hdr, int rowBegin, int rowEnd, IlmHandle albedo) const {
    IlmHDRConfiguration plain;
    if (!hdr) {      // hdr == null: InverseScaleFactor 1 and the material's default uniforms (exposure 1, gamma 1, offset 0)
        std::memset(&plain, 0, sizeof(plain));
        plain.Mode = ILM_HDR_NONE; plain.InverseScaleFactor = 1; plain.Exposure = 1; plain.Gamma = 1; plain.WhitePoint = 1;
        hdr = &plain;
    }
    if (rowEnd < 0) rowEnd = Configuration.RenderHeight;
    ThrowIfFailed(ilm_resolve_lighting_with_albedo(lightmap, albedo, destination, hdr, rowBegin, rowEnd));
}

void LightingRenderer::ReadLightmap(void* dst, int firstRow, int rowCount) const {
    ThrowIfFailed(ilm_lightmap_download(lightmap, dst, firstRow, rowCount));
}

}  // namespace Lighting
}  // namespace Illuminant
}  // namespace Squared

// bindings.cpp -- pybind11 glue exposing the C++ host mirror (illuminant_host.hpp) to Python as
// illuminant_amd._host, with the reference's class and member names.  Test / bench plumbing only:
// the product boundary is the C ABI underneath.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <cstring>

#include "illuminant_host.hpp"

namespace py = pybind11;
using namespace Squared::Illuminant;
using namespace Squared::Illuminant::Particles;
using namespace Squared::Illuminant::Particles::Transforms;
using namespace Squared::Illuminant::Lighting;

using farray = py::array_t<float, py::array::c_style | py::array::forcecast>;

static Vector2 v2(const std::vector<float>& v) { return Vector2{ v.at(0), v.at(1) }; }
static Vector3 v3(const std::vector<float>& v) { return Vector3{ v.at(0), v.at(1), v.at(2) }; }
static Vector4 v4(const std::vector<float>& v) { return Vector4{ v.at(0), v.at(1), v.at(2), v.at(3) }; }
static std::vector<float> l2(const Vector2& v) { return { v.X, v.Y }; }
static std::vector<float> l3(const Vector3& v) { return { v.X, v.Y, v.Z }; }
static std::vector<float> l4(const Vector4& v) { return { v.X, v.Y, v.Z, v.W }; }

// Vector-valued members are exposed as Python lists
#define VEC_PROP(cls, name, N) \
    .def_property(#name, [](const cls& o) { return l##N(o.name); }, [](cls& o, const std::vector<float>& v) { o.name = v##N(v); })

PYBIND11_MODULE(_host, m) {
    m.doc() = "C++ host mirror of Illuminant's ParticleSystem / LightingRenderer interface over libilluminant_hip.so";

    py::register_exception<InvalidOperationException>(m, "InvalidOperationException", PyExc_RuntimeError);
    py::register_exception<NativeException>(m, "NativeException", PyExc_RuntimeError);

    py::class_<Xoshiro>(m, "Xoshiro").def(py::init<uint64_t>(), py::arg("seed") = 0x1234567ull)
        .def("NextDouble", &Xoshiro::NextDouble).def("NextUInt64", &Xoshiro::NextUInt64);

    py::class_<ITimeProvider>(m, "ITimeProvider");
    py::class_<ManualTimeProvider, ITimeProvider>(m, "ManualTimeProvider").def(py::init<>())
        .def_readwrite("Now", &ManualTimeProvider::Now).def("Advance", &ManualTimeProvider::Advance).def("Seconds", &ManualTimeProvider::Seconds);

    py::class_<DeviceContext>(m, "DeviceContext").def(py::init<int>(), py::arg("deviceId") = 0)
        .def_static("FromHandle", [](IlmHandle member) { return new DeviceContext(DeviceContext::Borrowed{ member }); }, py::arg("memberContext"),
                    "wrap a context owned by a multi-device group (ilm_group_ctx); it is not destroyed with this object")
        .def("Sync", &DeviceContext::Sync).def("TimerStart", &DeviceContext::TimerStart).def("TimerStop", &DeviceContext::TimerStop)
        .def_property_readonly("Handle", &DeviceContext::Handle);

    py::class_<RenderTarget>(m, "RenderTarget")
        .def(py::init<DeviceContext&, int, int, int>(), py::arg("ctx"), py::arg("width"), py::arg("height"), py::arg("format") = (int)ILM_LIGHTMAP_FLOAT4,
             py::keep_alive<1, 2>())
        .def_readonly("Width", &RenderTarget::Width).def_readonly("Height", &RenderTarget::Height).def_readonly("Format", &RenderTarget::Format)
        .def("Clear", [](RenderTarget& t, const std::vector<float>& c) { t.Clear(Vector4{ c.at(0), c.at(1), c.at(2), c.at(3) }); })
        // (height, width, 4) float32 for a float4 target; raw bytes reshaped by the caller otherwise
        .def("Download", [](const RenderTarget& t) {
            if (t.Format != ILM_LIGHTMAP_FLOAT4) throw std::invalid_argument("Download(): float4 targets only");
            py::array_t<float> out({ (py::ssize_t)t.Height, (py::ssize_t)t.Width, (py::ssize_t)4 });
            t.Download(out.mutable_data());
            return out; });

    py::class_<DistanceField::Layout>(m, "DistanceFieldLayout")
        .def_readonly("Resolution", &DistanceField::Layout::Resolution)
        .def_readonly("SliceWidth", &DistanceField::Layout::SliceWidth).def_readonly("SliceHeight", &DistanceField::Layout::SliceHeight)
        .def_readonly("SliceCount", &DistanceField::Layout::SliceCount).def_readonly("PhysicalSliceCount", &DistanceField::Layout::PhysicalSliceCount)
        .def_readonly("ColumnCount", &DistanceField::Layout::ColumnCount).def_readonly("RowCount", &DistanceField::Layout::RowCount)
        .def_readonly("TextureWidth", &DistanceField::Layout::TextureWidth).def_readonly("TextureHeight", &DistanceField::Layout::TextureHeight);
    py::class_<DistanceField>(m, "DistanceField")
        .def_static("ComputeLayout", &DistanceField::ComputeLayout, py::arg("virtualWidth"), py::arg("virtualHeight"),
                    py::arg("requestedSliceCount"), py::arg("requestedResolution") = 1.0)
        .def(py::init<DeviceContext&, int, int, float, int, double, int, int>(), py::arg("ctx"), py::arg("virtualWidth"), py::arg("virtualHeight"),
             py::arg("virtualDepth"), py::arg("requestedSliceCount"), py::arg("requestedResolution") = 1.0,
             py::arg("maximumEncodedDistance") = 128, py::arg("format") = 0, py::keep_alive<1, 2>())
        .def_readonly("VirtualWidth", &DistanceField::VirtualWidth).def_readonly("VirtualHeight", &DistanceField::VirtualHeight)
        .def_readonly("VirtualDepth", &DistanceField::VirtualDepth).def_readonly("Resolution", &DistanceField::Resolution)
        .def_readonly("SliceWidth", &DistanceField::SliceWidth).def_readonly("SliceHeight", &DistanceField::SliceHeight)
        .def_readonly("SliceCount", &DistanceField::SliceCount).def_readonly("PhysicalSliceCount", &DistanceField::PhysicalSliceCount)
        .def_readonly("ColumnCount", &DistanceField::ColumnCount).def_readonly("RowCount", &DistanceField::RowCount)
        .def_readonly("TextureWidth", &DistanceField::TextureWidth).def_readonly("TextureHeight", &DistanceField::TextureHeight)
        .def_readwrite("ZOffset", &DistanceField::ZOffset)
        .def("Load", [](DistanceField& f, py::array_t<uint16_t, py::array::c_style | py::array::forcecast> a) {
            if (a.size() != (py::ssize_t)f.TextureWidth * f.TextureHeight * 4) throw ArgumentException("atlas size mismatch");
            f.Load(a.data());
        })
        .def("Save", [](const DistanceField& f) {
            py::array_t<uint16_t> out({ (py::ssize_t)f.TextureHeight, (py::ssize_t)f.TextureWidth, (py::ssize_t)4 });
            f.Save(out.mutable_data());
            return out;
        })
        .def("Invalidate", [](DistanceField& f) { f.Invalidate(); })
        .def_property_readonly("ValidSliceCount", [](const DistanceField& f) { return f.Slices.ValidSliceCount; })
        .def_property_readonly("InvalidSlices", [](const DistanceField& f) { return f.Slices.InvalidSlices; })
        .def_property_readonly("IsFullyGenerated", &DistanceField::IsFullyGenerated)
        .def_property_readonly("NeedsRasterize", &DistanceField::NeedsRasterize)
        .def_property_readonly("TextureHandle", &DistanceField::Texture)
        .def("ReadTexture", [](const DistanceField& f) {      // the atlas whatever its validity (tests)
            py::array_t<uint16_t> out({ (py::ssize_t)f.TextureHeight, (py::ssize_t)f.TextureWidth, (py::ssize_t)4 });
            ThrowIfFailed(ilm_sdf_download(f.Texture(), out.mutable_data()));
            return out;
        })
        .def("GetUniformsBytes", [](const DistanceField& f) { auto u = f.GetUniforms(); return py::bytes((const char*)&u, sizeof(u)); });
    py::class_<DynamicDistanceField, DistanceField>(m, "DynamicDistanceField")
        .def(py::init<DeviceContext&, int, int, float, int, double, int, int>(), py::arg("ctx"), py::arg("virtualWidth"), py::arg("virtualHeight"),
             py::arg("virtualDepth"), py::arg("sliceCount"), py::arg("requestedResolution") = 1.0,
             py::arg("maximumEncodedDistance") = 128, py::arg("format") = 0, py::keep_alive<1, 2>())
        .def("Invalidate", [](DynamicDistanceField& f, bool invalidateStatic) { f.Invalidate(invalidateStatic); }, py::arg("invalidateStatic") = true)
        .def_property_readonly("StaticValidSliceCount", [](const DynamicDistanceField& f) { return f.StaticSliceInfo.ValidSliceCount; })
        .def_property_readonly("StaticInvalidSlices", [](const DynamicDistanceField& f) { return f.StaticSliceInfo.InvalidSlices; })
        .def("ReadStaticTexture", [](const DynamicDistanceField& f) {
            py::array_t<uint16_t> out({ (py::ssize_t)f.TextureHeight, (py::ssize_t)f.TextureWidth, (py::ssize_t)4 });
            ThrowIfFailed(ilm_sdf_download(f.StaticTexture(), out.mutable_data()));
            return out;
        });

    // ---- particles -------------------------------------------------------------------------------------------
    py::class_<ParticleEngineConfiguration>(m, "ParticleEngineConfiguration").def(py::init<int>(), py::arg("chunkSize") = 256)
        .def_readwrite("ChunkSize", &ParticleEngineConfiguration::ChunkSize)
        .def_readwrite("TimeProvider", &ParticleEngineConfiguration::TimeProvider)
        .def_readwrite("UpdatesPerSecond", &ParticleEngineConfiguration::UpdatesPerSecond)
        .def_readwrite("MaximumUpdateDeltaTimeSeconds", &ParticleEngineConfiguration::MaximumUpdateDeltaTimeSeconds)
        .def_readwrite("AccurateLivenessCounts", &ParticleEngineConfiguration::AccurateLivenessCounts);

    py::class_<ParticleEngine>(m, "ParticleEngine")
        .def(py::init([](DeviceContext& ctx, const ParticleEngineConfiguration& cfg, farray rnd) {
            if (rnd.size() != (py::ssize_t)ParticleEngine::RandomnessTextureWidth * ParticleEngine::RandomnessTextureHeight * 4)
                throw ArgumentException("randomness table must be 653 x 807 x 4 floats");
            return new ParticleEngine(ctx, cfg, rnd.data());
        }), py::keep_alive<1, 2>())
        .def_readonly("Configuration", &ParticleEngine::Configuration);

    py::class_<BezierF>(m, "BezierF").def(py::init<>())
        .def_readwrite("Count", &BezierF::Count).def_readwrite("Mode", &BezierF::Mode)
        .def_readwrite("MinValue", &BezierF::MinValue).def_readwrite("MaxValue", &BezierF::MaxValue)
        .def_readwrite("A", &BezierF::A).def_readwrite("B", &BezierF::B).def_readwrite("C", &BezierF::C).def_readwrite("D", &BezierF::D);
    py::class_<Bezier4>(m, "Bezier4").def(py::init<>())
        .def_readwrite("Count", &Bezier4::Count).def_readwrite("Mode", &Bezier4::Mode)
        .def_readwrite("MinValue", &Bezier4::MinValue).def_readwrite("MaxValue", &Bezier4::MaxValue)
        VEC_PROP(Bezier4, A, 4) VEC_PROP(Bezier4, B, 4) VEC_PROP(Bezier4, C, 4) VEC_PROP(Bezier4, D, 4);

    py::class_<ParticleCollision>(m, "ParticleCollision").def(py::init<>())
        .def_readwrite("DistanceField", &ParticleCollision::Field)
        .def_readwrite("DistanceFieldMaximumZ", &ParticleCollision::DistanceFieldMaximumZ)
        .def_readwrite("Distance", &ParticleCollision::Distance).def_readwrite("LifePenalty", &ParticleCollision::LifePenalty)
        .def_readwrite("EscapeVelocity", &ParticleCollision::EscapeVelocity)
        .def_readwrite("BounceVelocityMultiplier", &ParticleCollision::BounceVelocityMultiplier);
    py::class_<ParticleColor>(m, "ParticleColor").def(py::init<>())
        VEC_PROP(ParticleColor, Global, 4)
        .def_readwrite("OpacityFromLife", &ParticleColor::OpacityFromLife)
        .def_readwrite("ColorFromLife", &ParticleColor::ColorFromLife).def_readwrite("ColorFromVelocity", &ParticleColor::ColorFromVelocity);
    py::class_<ParticleAppearance>(m, "ParticleAppearance").def(py::init<>())
        .def_property("TextureSize", [](const ParticleAppearance& a) -> py::object { if (!a.TextureSize) return py::none(); return py::cast(l2(*a.TextureSize)); },
                      [](ParticleAppearance& a, py::object v) { if (v.is_none()) a.TextureSize.reset(); else a.TextureSize = v2(v.cast<std::vector<float>>()); })
        VEC_PROP(ParticleAppearance, OffsetPx, 2)
        .def_property("SizePx", [](const ParticleAppearance& a) -> py::object { if (!a.SizePx) return py::none(); return py::cast(l2(*a.SizePx)); },
                      [](ParticleAppearance& a, py::object v) { if (v.is_none()) a.SizePx.reset(); else a.SizePx = v2(v.cast<std::vector<float>>()); })
        VEC_PROP(ParticleAppearance, AnimationRate, 2)
        .def_readwrite("RelativeSize", &ParticleAppearance::RelativeSize)
        .def_readwrite("ColumnFromVelocity", &ParticleAppearance::ColumnFromVelocity).def_readwrite("RowFromVelocity", &ParticleAppearance::RowFromVelocity)
        .def_readwrite("Rounded", &ParticleAppearance::Rounded).def_readwrite("DitheredOpacity", &ParticleAppearance::DitheredOpacity)
        .def_readwrite("Bilinear", &ParticleAppearance::Bilinear)
        .def_readwrite("RoundingPowerFromLife", &ParticleAppearance::RoundingPowerFromLife);
    py::class_<ParticleSystemConfiguration>(m, "ParticleSystemConfiguration").def(py::init<>())
        .def_readwrite("Appearance", &ParticleSystemConfiguration::Appearance)
        .def_readwrite("AutoReadback", &ParticleSystemConfiguration::AutoReadback).def_readwrite("SortedReadback", &ParticleSystemConfiguration::SortedReadback)
        VEC_PROP(ParticleSystemConfiguration, Size, 2)
        .def_readwrite("Friction", &ParticleSystemConfiguration::Friction)
        .def_readwrite("MaximumVelocity", &ParticleSystemConfiguration::MaximumVelocity)
        .def_readwrite("LifeDecayPerSecond", &ParticleSystemConfiguration::LifeDecayPerSecond)
        .def_readwrite("Collision", &ParticleSystemConfiguration::Collision)
        .def_readwrite("Color", &ParticleSystemConfiguration::Color)
        .def_readwrite("SizeFromLife", &ParticleSystemConfiguration::SizeFromLife)
        .def_readwrite("SizeFromVelocity", &ParticleSystemConfiguration::SizeFromVelocity)
        .def_readwrite("RotationFromLife", &ParticleSystemConfiguration::RotationFromLife)
        .def_readwrite("RotationFromIndex", &ParticleSystemConfiguration::RotationFromIndex)
        .def_readwrite("RotationFromVelocity", &ParticleSystemConfiguration::RotationFromVelocity)
        .def_readwrite("ZToY", &ParticleSystemConfiguration::ZToY)
        .def_readwrite("StippleFactor", &ParticleSystemConfiguration::StippleFactor)
        VEC_PROP(ParticleSystemConfiguration, ZFormula, 4)
        .def_readwrite("SizeFromZ", &ParticleSystemConfiguration::SizeFromZ)
        .def_readwrite("TimeProvider", &ParticleSystemConfiguration::TimeProvider);

    py::enum_<AreaType>(m, "AreaType").value("None_", AreaType::None).value("Ellipsoid", AreaType::Ellipsoid).value("Box", AreaType::Box)
        .value("Cylinder", AreaType::Cylinder).value("Spheroid", AreaType::Spheroid).value("Octagon", AreaType::Octagon);
    py::class_<TransformArea>(m, "TransformArea").def(py::init<>())
        .def_readwrite("Type", &TransformArea::Type) VEC_PROP(TransformArea, Center, 3) VEC_PROP(TransformArea, Size, 3)
        .def_readwrite("Falloff", &TransformArea::Falloff).def_readwrite("Rotation", &TransformArea::Rotation);

    py::class_<ParticleTransform>(m, "ParticleTransform")
        .def_readwrite("IsActive", &ParticleTransform::IsActive).def_readwrite("IsActive2", &ParticleTransform::IsActive2)
        .def_readwrite("Label", &ParticleTransform::Label).def_property_readonly("IsValid", &ParticleTransform::IsValid)
        .def("Reset", &ParticleTransform::Reset);
    py::class_<ParticleAreaTransform, ParticleTransform>(m, "ParticleAreaTransform")
        .def_readwrite("Strength", &ParticleAreaTransform::Strength)
        .def_property("CategoryFilter", [](const ParticleAreaTransform& t) -> py::object { if (!t.CategoryFilter) return py::none(); return py::cast(l2(*t.CategoryFilter)); },
                      [](ParticleAreaTransform& t, py::object v) { if (v.is_none()) t.CategoryFilter.reset(); else t.CategoryFilter = v2(v.cast<std::vector<float>>()); })
        .def_readwrite("Area", &ParticleAreaTransform::Area);

    py::class_<FMA::FMAParameters>(m, "FMAParameters").def(py::init<>()) VEC_PROP(FMA::FMAParameters, Add, 3) VEC_PROP(FMA::FMAParameters, Multiply, 3);
    py::class_<FMA, ParticleAreaTransform>(m, "FMA").def(py::init<>())
        .def_readwrite("CyclesPerSecond", &FMA::CyclesPerSecond)
        .def_readwrite("Position", &FMA::Position).def_readwrite("Velocity", &FMA::Velocity);

    py::class_<Noise::P4>(m, "NoiseParameters4").def(py::init<>()) VEC_PROP(Noise::P4, Offset, 4) VEC_PROP(Noise::P4, Minimum, 4) VEC_PROP(Noise::P4, Scale, 4);
    py::class_<Noise::P3>(m, "NoiseParameters3").def(py::init<>()) VEC_PROP(Noise::P3, Offset, 3) VEC_PROP(Noise::P3, Minimum, 3) VEC_PROP(Noise::P3, Scale, 3);
    py::class_<Noise::PF>(m, "NoiseParametersF").def(py::init<>())
        .def_readwrite("Offset", &Noise::PF::Offset).def_readwrite("Minimum", &Noise::PF::Minimum).def_readwrite("Scale", &Noise::PF::Scale);
    py::class_<Noise, ParticleAreaTransform>(m, "Noise").def(py::init<uint64_t>(), py::arg("seed") = 1)
        .def_readwrite("CyclesPerSecond", &Noise::CyclesPerSecond)
        .def_readwrite("Position", &Noise::Position).def_readwrite("Velocity", &Noise::Velocity).def_readwrite("Speed", &Noise::Speed)
        .def_readwrite("Interval", &Noise::Interval).def_readwrite("ReplaceOldVelocity", &Noise::ReplaceOldVelocity)
        .def_readonly("CurrentU", &Noise::CurrentU).def_readonly("CurrentV", &Noise::CurrentV)
        .def_readonly("NextU", &Noise::NextU).def_readonly("NextV", &Noise::NextV);

    py::class_<SpatialNoise, Noise>(m, "SpatialNoise").def(py::init<uint64_t>(), py::arg("seed") = 1)
        VEC_PROP(SpatialNoise, SpaceScale, 2);
    py::class_<MatrixMultiply, ParticleAreaTransform>(m, "MatrixMultiply").def(py::init<>())
        .def_readwrite("CyclesPerSecond", &MatrixMultiply::CyclesPerSecond)
        .def_property("Position", [](const MatrixMultiply& t) { return std::vector<float>(t.Position.m, t.Position.m + 16); },
                      [](MatrixMultiply& t, const std::vector<float>& v) { for (int i = 0; i < 16; i++) t.Position.m[i] = v.at((size_t)i); })
        .def_property("Velocity", [](const MatrixMultiply& t) { return std::vector<float>(t.Velocity.m, t.Velocity.m + 16); },
                      [](MatrixMultiply& t, const std::vector<float>& v) { for (int i = 0; i < 16; i++) t.Velocity.m[i] = v.at((size_t)i); });

    py::enum_<AttractorType>(m, "AttractorType").value("Physical", AttractorType::Physical).value("Linear", AttractorType::Linear)
        .value("Exponential", AttractorType::Exponential);
    py::class_<Gravity::Attractor>(m, "Attractor").def(py::init<>())
        VEC_PROP(Gravity::Attractor, Position, 3)
        .def_readwrite("Radius", &Gravity::Attractor::Radius).def_readwrite("Strength", &Gravity::Attractor::Strength)
        .def_readwrite("Type", &Gravity::Attractor::Type);
    py::class_<Gravity, ParticleTransform>(m, "Gravity").def(py::init<>())
        .def_readwrite("MaximumAcceleration", &Gravity::MaximumAcceleration)
        .def_readwrite("Attractors", &Gravity::Attractors)
        VEC_PROP(Gravity, CategoryFilter, 2);

    py::enum_<FormulaType>(m, "FormulaType").value("Linear", FormulaType::Linear).value("Spherical", FormulaType::Spherical)
        .value("Towards", FormulaType::Towards).value("Rectangular", FormulaType::Rectangular);
    py::class_<Formula1>(m, "Formula1").def(py::init<>())
        .def_readwrite("Constant", &Formula1::Constant).def_readwrite("RandomScale", &Formula1::RandomScale).def_readwrite("Offset", &Formula1::Offset);
    py::class_<Formula3>(m, "Formula3").def(py::init<>())
        VEC_PROP(Formula3, Constant, 3) VEC_PROP(Formula3, RandomScale, 3) VEC_PROP(Formula3, Offset, 3)
        .def_readwrite("Type", &Formula3::Type);
    py::class_<Formula4>(m, "Formula4").def(py::init<>())
        VEC_PROP(Formula4, Constant, 4) VEC_PROP(Formula4, RandomScale, 4) VEC_PROP(Formula4, Offset, 4);

    py::class_<SpawnerBase, ParticleTransform>(m, "SpawnerBase")
        .def_readwrite("MinRate", &SpawnerBase::MinRate).def_readwrite("MaxRate", &SpawnerBase::MaxRate)
        .def_readwrite("MaximumTotal", &SpawnerBase::MaximumTotal)
        .def_readwrite("Position", &SpawnerBase::Position).def_readwrite("Velocity", &SpawnerBase::Velocity)
        .def_readwrite("Life", &SpawnerBase::Life).def_readwrite("Category", &SpawnerBase::Category).def_readwrite("Color", &SpawnerBase::Color)
        .def_readwrite("AlignVelocityAndPosition", &SpawnerBase::AlignVelocityAndPosition)
        VEC_PROP(SpawnerBase, AxisMask, 3)
        .def_readwrite("AlphaDiscardThreshold", &SpawnerBase::AlphaDiscardThreshold)
        .def_readwrite("RateError", &SpawnerBase::RateError)
        .def_readwrite("ScriptedDraws", &SpawnerBase::ScriptedDraws)
        .def_property_readonly("TotalSpawned", &SpawnerBase::TotalSpawned)
        .def("BeginTick", [](SpawnerBase& s, double now, double dt) { int n = 0; s.BeginTick(now, dt, n); return n; })
        .def("EndTick", &SpawnerBase::EndTick);
    py::class_<Spawner, SpawnerBase>(m, "Spawner").def(py::init<uint64_t>(), py::arg("seed") = 1)
        .def_property("AdditionalPositions",
                      [](const Spawner& s) { std::vector<std::vector<float>> r; for (auto& p : s.AdditionalPositions) r.push_back(l3(p)); return r; },
                      [](Spawner& s, const std::vector<std::vector<float>>& v) { s.AdditionalPositions.clear(); for (auto& p : v) s.AdditionalPositions.push_back(v3(p)); })
        .def_readwrite("PolygonRate", &Spawner::PolygonRate).def_readwrite("PolygonLoop", &Spawner::PolygonLoop)
        .def_readwrite("VelocityAlongPolygon", &Spawner::VelocityAlongPolygon).def_readwrite("RatePerPosition", &Spawner::RatePerPosition);

    py::class_<FeedbackSpawner, SpawnerBase>(m, "FeedbackSpawner").def(py::init<uint64_t>(), py::arg("seed") = 1)
        .def_property("SourceSystem", py::cpp_function([](FeedbackSpawner& s) { return s.SourceSystem; }, py::return_value_policy::reference),
                      py::cpp_function([](FeedbackSpawner& s, ParticleSystem* p) { s.SourceSystem = p; }, py::keep_alive<1, 2>()))
        .def_readwrite("SlidingWindowSize", &FeedbackSpawner::SlidingWindowSize).def_readwrite("SlidingWindowMargin", &FeedbackSpawner::SlidingWindowMargin)
        .def_readwrite("SpawnFromEntireWindow", &FeedbackSpawner::SpawnFromEntireWindow)
        .def_readwrite("InstanceMultiplier", &FeedbackSpawner::InstanceMultiplier)
        .def_readwrite("AlignPositionConstant", &FeedbackSpawner::AlignPositionConstant)
        .def_readwrite("SourceVelocityFactor", &FeedbackSpawner::SourceVelocityFactor)
        .def_readwrite("MultiplyLife", &FeedbackSpawner::MultiplyLife).def_readwrite("MultiplyColorConstant", &FeedbackSpawner::MultiplyColorConstant)
        VEC_PROP(FeedbackSpawner, SourceLifeRange, 2);

    py::class_<PatternSpawner, SpawnerBase>(m, "PatternSpawner").def(py::init<uint64_t>(), py::arg("seed") = 1)
        .def_property("TextureTopLeftPx",
                      [](const PatternSpawner& s) { return s.TextureTopLeftPx ? std::optional<std::vector<float>>(l2(*s.TextureTopLeftPx)) : std::nullopt; },
                      [](PatternSpawner& s, const std::optional<std::vector<float>>& v) { s.TextureTopLeftPx = v ? std::optional<Vector2>(v2(*v)) : std::nullopt; })
        .def_property("TextureSizePx",
                      [](const PatternSpawner& s) { return s.TextureSizePx ? std::optional<std::vector<float>>(l2(*s.TextureSizePx)) : std::nullopt; },
                      [](PatternSpawner& s, const std::optional<std::vector<float>>& v) { s.TextureSizePx = v ? std::optional<Vector2>(v2(*v)) : std::nullopt; })
        .def_readwrite("MipBiasBase", &PatternSpawner::MipBiasBase).def_readwrite("WholeSpawn", &PatternSpawner::WholeSpawn)
        .def_readwrite("InstantInitialSpawn", &PatternSpawner::InstantInitialSpawn)
        .def_readwrite("MultiplyColorConstant", &PatternSpawner::MultiplyColorConstant)
        .def_property("Divisor", &PatternSpawner::Divisor, &PatternSpawner::SetDivisor)
        .def_property_readonly("ParticlesPerRow", &PatternSpawner::ParticlesPerRow)
        .def_property_readonly("RowsPerInstance", &PatternSpawner::RowsPerInstance)
        .def_property_readonly("ParticlesPerInstance", &PatternSpawner::ParticlesPerInstance)
        .def_property_readonly("RowsSpawned", &PatternSpawner::RowsSpawned)
        // levels: list of (h, w, 4) float32 arrays, level 0 first
        .def("SetTexture", [](PatternSpawner& s, const std::vector<py::array_t<float, py::array::c_style | py::array::forcecast>>& levels) {
            if (levels.empty()) { s.SetTexture(0, 0, 0, {}); return; }
            std::vector<IlmFloat4> flat;
            for (const auto& a : levels) {
                if (a.ndim() != 3 || a.shape(2) != 4) throw std::invalid_argument("each level must be (h, w, 4) float32");
                const IlmFloat4* p = reinterpret_cast<const IlmFloat4*>(a.data());
                flat.insert(flat.end(), p, p + a.shape(0) * a.shape(1));
            }
            s.SetTexture((int)levels[0].shape(1), (int)levels[0].shape(0), (int)levels.size(), std::move(flat));
        });

    py::class_<ParticleSystem::Chunk>(m, "Chunk")
        .def_readonly("IsFeedbackSource", &ParticleSystem::Chunk::IsFeedbackSource)
        .def_readonly("TotalConsumedForFeedback", &ParticleSystem::Chunk::TotalConsumedForFeedback)
        .def_readonly("ID", &ParticleSystem::Chunk::ID).def_readonly("NextSpawnOffset", &ParticleSystem::Chunk::NextSpawnOffset)
        .def_readonly("TotalSpawned", &ParticleSystem::Chunk::TotalSpawned)
        .def_readonly("NoLongerASpawnTarget", &ParticleSystem::Chunk::NoLongerASpawnTarget)
        .def_readonly("Count", &ParticleSystem::Chunk::Count).def_readonly("DeadFrameCount", &ParticleSystem::Chunk::DeadFrameCount);
    py::class_<ParticleSystem::UpdateResult>(m, "UpdateResult")
        .def_readonly("PerformedUpdate", &ParticleSystem::UpdateResult::PerformedUpdate)
        .def_readonly("Timestamp", &ParticleSystem::UpdateResult::Timestamp);
    py::class_<ParticleSystem>(m, "ParticleSystem")
        .def(py::init<ParticleEngine&, const ParticleSystemConfiguration&>(), py::keep_alive<1, 2>())
        .def_readwrite("Configuration", &ParticleSystem::Configuration)
        .def_readonly("LiveCount", &ParticleSystem::LiveCount)
        .def_property_readonly("Capacity", &ParticleSystem::Capacity)
        .def_property_readonly("Chunks", &ParticleSystem::Chunks)
        .def_readonly("TotalSpawnCount", &ParticleSystem::TotalSpawnCount)
        .def_readwrite("DeadFrameThreshold", &ParticleSystem::DeadFrameThreshold)
        .def_readwrite("BlockingLivenessReadback", &ParticleSystem::BlockingLivenessReadback)
        .def_readonly("LastDeltaTimeSeconds", &ParticleSystem::LastDeltaTimeSeconds)
        // Transforms list: the Python side keeps the objects alive (AddTransform keeps a reference)
        .def("AddTransform", [](ParticleSystem& s, ParticleTransform* t) { s.Transforms.push_back(t); }, py::keep_alive<1, 2>())
        .def("ClearTransforms", [](ParticleSystem& s) { s.Transforms.clear(); })
        .def("Spawn", [](ParticleSystem& s, int count, farray pos, farray vel, py::object color) {
            if (pos.size() < (py::ssize_t)count * 4 || vel.size() < (py::ssize_t)count * 4) throw ArgumentException("initializer arrays too small");
            const IlmFloat4* c = nullptr;
            farray carr;
            if (!color.is_none()) { carr = color.cast<farray>(); c = (const IlmFloat4*)carr.data(); }
            return s.Spawn(count, (const IlmFloat4*)pos.data(), (const IlmFloat4*)vel.data(), c);
        }, py::arg("particleCount"), py::arg("positions"), py::arg("velocities"), py::arg("colors") = py::none())
        .def("Update", &ParticleSystem::Update, py::arg("frameIndex"))
        .def("UpdateMany", [](ParticleSystem& s, ManualTimeProvider& tp, int firstFrame, int count, double dt) {
            // `count` Update calls, advancing the manual clock by dt before each (bench inner loop without Python in it)
            for (int i = 0; i < count; i++) { tp.Advance(dt); s.Update(firstFrame + i); }
        })
        .def("Clear", &ParticleSystem::Clear)
        // Render(target, blendMode, origin, scale, viewportScale, viewportPosition, wantStats) -> (live quads, tile pairs, shaded pixels)
        .def("Render", [](const ParticleSystem& s, RenderTarget& target, int blendMode, const std::vector<float>& origin, const std::vector<float>& scale,
                          const std::vector<float>& viewportScale, const std::vector<float>& viewportPosition, bool wantStats) {
            ParticleSystem::RenderParameters rp;
            rp.Origin = v2(origin); rp.Scale = v2(scale);
            const ParticleSystem::RenderStats st = s.Render(target, blendMode, &rp, v2(viewportScale), v2(viewportPosition), wantStats);
            return py::make_tuple(st.LiveQuads, st.TilePairs, st.ShadedPixels);
        }, py::arg("target"), py::arg("blendMode") = (int)ILM_BLEND_ALPHA, py::arg("origin") = std::vector<float>{0, 0},
           py::arg("scale") = std::vector<float>{1, 1}, py::arg("viewportScale") = std::vector<float>{1, 1},
           py::arg("viewportPosition") = std::vector<float>{0, 0}, py::arg("wantStats") = false)
        // SetBitmap((h, w, 4) float32) / SetBitmap(None)
        .def("SetBitmap", [](ParticleSystem& s, py::object texels) {
            if (texels.is_none()) { s.SetBitmap(0, 0, nullptr); return; }
            auto a = py::array_t<float, py::array::c_style | py::array::forcecast>::ensure(texels);
            if (!a || a.ndim() != 3 || a.shape(2) != 4) throw std::invalid_argument("bitmap must be (h, w, 4) float32");
            s.SetBitmap((int)a.shape(1), (int)a.shape(0), reinterpret_cast<const IlmFloat4*>(a.data())); })
        .def("RasterizeParamsBytes", [](const ParticleSystem& s, int blendMode, const std::vector<float>& origin, const std::vector<float>& scale,
                                        const std::vector<float>& viewportScale, const std::vector<float>& viewportPosition) {
            ParticleSystem::RenderParameters rp;
            rp.Origin = v2(origin); rp.Scale = v2(scale);
            const IlmRasterizeParams p = s.GetRasterizeParams(blendMode, &rp, v2(viewportScale), v2(viewportPosition));
            return py::bytes((const char*)&p, sizeof(p)); })
        // (n, 12) float32 array over the pinned read-back buffer: no copy; valid until the next read-back on the context
        .def("PerformReadbackView", [](const ParticleSystem& s) {
            const ParticleSystem::ReadbackView v = s.PerformReadbackView();
            constexpr py::ssize_t kFloats = sizeof(IlmReadbackDrawCall) / sizeof(float);
            static const float kEmpty = 0.0f;
            return py::array_t<float>({ (py::ssize_t)v.Count, kFloats }, { (py::ssize_t)sizeof(IlmReadbackDrawCall), (py::ssize_t)sizeof(float) },
                                      v.Count ? reinterpret_cast<const float*>(v.Records) : &kEmpty, py::none()); })
        .def("PerformReadback", [](const ParticleSystem& s) {
            auto r = s.PerformReadback();
            return py::bytes((const char*)r.data(), r.size() * sizeof(IlmReadbackDrawCall)); })
        .def_property_readonly("ReadbackResultBytes", [](const ParticleSystem& s) {
            return py::bytes((const char*)s.ReadbackResult.data(), s.ReadbackResult.size() * sizeof(IlmReadbackDrawCall)); })
        .def("GetReadbackParamsBytes", [](const ParticleSystem& s) { auto p = s.GetReadbackParams(); return py::bytes((const char*)&p, sizeof(p)); })
        .def("Readback", [](const ParticleSystem& s, int chunk, int plane) {
            farray out({ (py::ssize_t)s.ChunkMaximumCount(), (py::ssize_t)4 });
            s.Readback(chunk, plane, (IlmFloat4*)out.mutable_data());
            return out;
        })
        .def_property_readonly("Handle", &ParticleSystem::Handle)
        .def("LastStepBytes", [](const ParticleSystem& s) { return py::bytes((const char*)&s.LastStep(), sizeof(IlmStepDesc)); });

    // ---- lighting ------------------------------------------------------------------------------------------------
    // RampTexture((h, w, 4) float32)
    py::class_<RampTexture, std::shared_ptr<RampTexture>>(m, "RampTexture")
        .def(py::init([](py::array_t<float, py::array::c_style | py::array::forcecast> a) {
            if (a.ndim() != 3 || a.shape(2) != 4) throw std::invalid_argument("ramp texture must be (h, w, 4) float32");
            auto t = std::make_shared<RampTexture>();
            t->Width = (int)a.shape(1); t->Height = (int)a.shape(0);
            const IlmFloat4* p = reinterpret_cast<const IlmFloat4*>(a.data());
            t->Texels.assign(p, p + a.shape(0) * a.shape(1));
            return t; }))
        .def_readonly("Width", &RampTexture::Width).def_readonly("Height", &RampTexture::Height);
    py::class_<SphereLightSource>(m, "SphereLightSource").def(py::init<>())
        .def_readwrite("SortKey", &SphereLightSource::SortKey).def_readwrite("Enabled", &SphereLightSource::Enabled)
        VEC_PROP(SphereLightSource, Position, 3)
        .def_readwrite("Radius", &SphereLightSource::Radius).def_readwrite("RampLength", &SphereLightSource::RampLength)
        VEC_PROP(SphereLightSource, Color, 4)
        .def_readwrite("Opacity", &SphereLightSource::Opacity)
        .def_property("RampMode", [](const SphereLightSource& l) { return (int)l.RampMode; }, [](SphereLightSource& l, int v) { l.RampMode = (LightSourceRampMode)v; })
        .def_readwrite("CastsShadows", &SphereLightSource::CastsShadows)
        .def_readwrite("AmbientOcclusionRadius", &SphereLightSource::AmbientOcclusionRadius)
        .def_readwrite("AmbientOcclusionOpacity", &SphereLightSource::AmbientOcclusionOpacity)
        .def_readwrite("ShadowDistanceFalloff", &SphereLightSource::ShadowDistanceFalloff)
        .def_readwrite("FalloffYFactor", &SphereLightSource::FalloffYFactor)
        .def_readwrite("ShadowFilter", &SphereLightSource::ShadowFilter)
        VEC_PROP(SphereLightSource, SpecularColor, 3)
        .def_readwrite("SpecularPower", &SphereLightSource::SpecularPower)
        .def_readwrite("TextureRef", &SphereLightSource::TextureRef)
        .def_readwrite("Quality", &SphereLightSource::Quality)
        .def_readwrite("RampOffset", &SphereLightSource::RampOffset).def_readwrite("RampRate", &SphereLightSource::RampRate);
    py::class_<ReplicatedLight>(m, "ReplicatedLight").def(py::init<>())
        VEC_PROP(ReplicatedLight, Position, 3)
        .def_readwrite("Radius", &ReplicatedLight::Radius).def_readwrite("RampLength", &ReplicatedLight::RampLength)
        .def_readwrite("SpecularPower", &ReplicatedLight::SpecularPower).def_readwrite("Opacity", &ReplicatedLight::Opacity)
        .def_property("Color", [](const ReplicatedLight& l) -> py::object { if (!l.Color) return py::none(); return py::make_tuple(l.Color->X, l.Color->Y, l.Color->Z, l.Color->W); },
                      [](ReplicatedLight& l, py::object v) { if (v.is_none()) { l.Color.reset(); return; } auto t = v.cast<std::vector<float>>(); l.Color = Vector4{t.at(0), t.at(1), t.at(2), t.at(3)}; })
        .def_property("SpecularColor", [](const ReplicatedLight& l) -> py::object { if (!l.SpecularColor) return py::none(); return py::make_tuple(l.SpecularColor->X, l.SpecularColor->Y, l.SpecularColor->Z); },
                      [](ReplicatedLight& l, py::object v) { if (v.is_none()) { l.SpecularColor.reset(); return; } auto t = v.cast<std::vector<float>>(); l.SpecularColor = Vector3{t.at(0), t.at(1), t.at(2)}; });
    py::class_<LightSourceReplicator>(m, "LightSourceReplicator").def(py::init<>())
        .def_readwrite("SortKey", &LightSourceReplicator::SortKey).def_readwrite("Enabled", &LightSourceReplicator::Enabled)
        .def_readwrite("Template", &LightSourceReplicator::Template)
        .def_readwrite("Lights", &LightSourceReplicator::Lights)
        .def("Clear", &LightSourceReplicator::Clear).def("Add", &LightSourceReplicator::Add);
    py::class_<ParticleLightSource>(m, "ParticleLightSource").def(py::init<>())
        .def_readwrite("Template", &ParticleLightSource::Template)
        .def_property("System", py::cpp_function([](ParticleLightSource& s) { return s.System; }, py::return_value_policy::reference),
                      py::cpp_function([](ParticleLightSource& s, ParticleSystem* p) { s.System = p; }, py::keep_alive<1, 2>()))
        .def_readwrite("IsActive", &ParticleLightSource::IsActive).def_readwrite("Enabled", &ParticleLightSource::Enabled)
        .def_readwrite("StippleFactor", &ParticleLightSource::StippleFactor);
    py::class_<LightProbe, std::shared_ptr<LightProbe>>(m, "LightProbe").def(py::init<>())
        VEC_PROP(LightProbe, Position, 3)
        .def_property("Normal", [](const LightProbe& p) -> py::object { if (!p.Normal) return py::none(); return py::cast(l3(*p.Normal)); },
                      [](LightProbe& p, py::object v) { if (v.is_none()) p.Normal.reset(); else p.Normal = v3(v.cast<std::vector<float>>()); })
        .def_readwrite("EnableShadows", &LightProbe::EnableShadows)
        .def_property_readonly("Value", [](const LightProbe& p) { return l4(p.Value); })
        .def_property_readonly("PreviousValue", [](const LightProbe& p) { return l4(p.PreviousValue); });
    py::class_<LightProbeCollection>(m, "LightProbeCollection")
        .def("Add", &LightProbeCollection::Add).def("Clear", &LightProbeCollection::Clear)
        .def_property_readonly("Count", &LightProbeCollection::Count).def("__len__", &LightProbeCollection::Count)
        .def("__getitem__", [](LightProbeCollection& c, int i) { return c.Items.at((size_t)i); });
    py::class_<LightObstruction, std::shared_ptr<LightObstruction>>(m, "LightObstruction")
        .def(py::init([](int type, const std::vector<float>& center, const std::vector<float>& radius, float rotation) {
            return std::make_shared<LightObstruction>((LightObstructionType)type, v3(center), v3(radius), rotation);
        }), py::arg("type"), py::arg("center") = std::vector<float>{0, 0, 0}, py::arg("radius") = std::vector<float>{0, 0, 0}, py::arg("rotation") = 0.0f)
        .def_property("Type", [](const LightObstruction& o) { return (int)o.Type(); }, [](LightObstruction& o, int t) { o.SetType((LightObstructionType)t); })
        .def_property("Center", [](const LightObstruction& o) { return l3(o.Center()); }, [](LightObstruction& o, const std::vector<float>& v) { o.SetCenter(v3(v)); })
        .def_property("Size", [](const LightObstruction& o) { return l3(o.Size()); }, [](LightObstruction& o, const std::vector<float>& v) { o.SetSize(v3(v)); })
        .def_property("Orientation", [](const LightObstruction& o) { return l4(o.Orientation()); }, [](LightObstruction& o, const std::vector<float>& v) { o.SetOrientation(v4(v)); })
        .def_property("Rotation", &LightObstruction::Rotation, &LightObstruction::SetRotation)
        .def_property("IsDynamic", &LightObstruction::IsDynamic, &LightObstruction::SetIsDynamic)
        .def_readonly("IsValid", &LightObstruction::IsValid)
        .def("VertexBytes", [](const LightObstruction& o) { auto v = o.Vertex(); return py::bytes((const char*)&v, sizeof(v)); });
    py::class_<LightObstructionCollection>(m, "LightObstructionCollection")
        .def("Add", &LightObstructionCollection::Add).def("RemoveAt", &LightObstructionCollection::RemoveAt)
        .def("Clear", &LightObstructionCollection::Clear).def_property_readonly("Count", &LightObstructionCollection::Count)
        .def("__len__", &LightObstructionCollection::Count)
        .def("__getitem__", [](LightObstructionCollection& c, int i) { return c.Items.at(i); });
    py::class_<HeightVolume>(m, "HeightVolume").def(py::init<>())
        .def_property("Polygon", [](const HeightVolume& h) { std::vector<std::vector<float>> r; for (auto& p : h.Polygon) r.push_back(l2(p)); return r; },
                      [](HeightVolume& h, const std::vector<std::vector<float>>& v) { h.Polygon.clear(); for (auto& p : v) h.Polygon.push_back(v2(p)); })
        .def_readwrite("ZBase", &HeightVolume::ZBase).def_readwrite("Height", &HeightVolume::Height)
        .def_readwrite("IsDynamic", &HeightVolume::IsDynamic).def_readwrite("IsObstruction", &HeightVolume::IsObstruction)
        .def_readwrite("TopFaceEnableShadows", &HeightVolume::TopFaceEnableShadows);
    py::class_<LightingEnvironment>(m, "LightingEnvironment").def(py::init<>())
        .def_readwrite("Lights", &LightingEnvironment::Lights)
        .def_readwrite("Replicators", &LightingEnvironment::Replicators)
        .def_readwrite("ParticleLights", &LightingEnvironment::ParticleLights)
        .def_property_readonly("Obstructions", [](LightingEnvironment& e) -> LightObstructionCollection& { return e.Obstructions; }, py::return_value_policy::reference_internal)
        .def_readwrite("HeightVolumes", &LightingEnvironment::HeightVolumes)
        .def_readwrite("GroundZ", &LightingEnvironment::GroundZ).def_readwrite("MaximumZ", &LightingEnvironment::MaximumZ)
        .def_readwrite("ZToYMultiplier", &LightingEnvironment::ZToYMultiplier)
        .def_readwrite("EnableGroundShadows", &LightingEnvironment::EnableGroundShadows)
        VEC_PROP(LightingEnvironment, Ambient, 4);
    py::class_<RendererQualitySettings, std::shared_ptr<RendererQualitySettings>>(m, "RendererQualitySettings").def(py::init<>())
        .def_readwrite("MinStepSize", &RendererQualitySettings::MinStepSize).def_readwrite("LongStepFactor", &RendererQualitySettings::LongStepFactor)
        .def_readwrite("MaxStepCount", &RendererQualitySettings::MaxStepCount).def_readwrite("MaxConeRadius", &RendererQualitySettings::MaxConeRadius)
        .def_readwrite("ConeGrowthFactor", &RendererQualitySettings::ConeGrowthFactor)
        .def_readwrite("OcclusionToOpacityPower", &RendererQualitySettings::OcclusionToOpacityPower);
    py::class_<RendererConfiguration>(m, "RendererConfiguration").def(py::init<int, int>())
        .def_readwrite("RenderWidth", &RendererConfiguration::RenderWidth).def_readwrite("RenderHeight", &RendererConfiguration::RenderHeight)
        .def_readwrite("HighQuality", &RendererConfiguration::HighQuality).def_readwrite("TwoPointFiveD", &RendererConfiguration::TwoPointFiveD)
        .def_readwrite("LightOcclusion", &RendererConfiguration::LightOcclusion)
        VEC_PROP(RendererConfiguration, RenderScale, 2)
        .def_readwrite("DefaultQuality", &RendererConfiguration::DefaultQuality)
        .def_readwrite("MaximumFieldUpdatesPerFrame", &RendererConfiguration::MaximumFieldUpdatesPerFrame)
        .def_readwrite("DefaultRampTexture", &RendererConfiguration::DefaultRampTexture)
        .def_readwrite("MaximumLightProbeCount", &RendererConfiguration::MaximumLightProbeCount)
        .def_readwrite("EnableGBuffer", &RendererConfiguration::EnableGBuffer).def_readwrite("RenderGroundPlane", &RendererConfiguration::RenderGroundPlane)
        .def_readwrite("HighQualityGBuffer", &RendererConfiguration::HighQualityGBuffer)
        .def_readwrite("FloatLightmap", &RendererConfiguration::FloatLightmap);
    py::class_<LightingRenderer>(m, "LightingRenderer")
        .def(py::init([](DeviceContext& ctx, const RendererConfiguration& cfg, LightingEnvironment* env, uintptr_t externalLightmap) {
            return new LightingRenderer(ctx, cfg, env, reinterpret_cast<void*>(externalLightmap));
        }), py::arg("ctx"), py::arg("configuration"), py::arg("environment"), py::arg("externalLightmap") = 0,
             py::keep_alive<1, 2>(), py::keep_alive<1, 4>())
        .def_readwrite("Configuration", &LightingRenderer::Configuration)
        .def_property("DistanceField", py::cpp_function([](LightingRenderer& r) { return r.Field; }, py::return_value_policy::reference),
                      py::cpp_function([](LightingRenderer& r, DistanceField* f) { r.Field = f; }, py::keep_alive<1, 2>()))
        .def("SetGBuffer", [](LightingRenderer& r, py::object arr, int format) {
            if (arr.is_none()) { r.SetGBuffer(nullptr, 0, 0, 0); return; }
            py::array a = arr.cast<py::array>();
            if (a.ndim() != 3 || a.shape(2) != 4) throw ArgumentException("G-buffer must be (H, W, 4)");
            r.SetGBuffer(a.data(), (int)a.shape(1), (int)a.shape(0), format);
        })
        .def("UpdateFields", &LightingRenderer::UpdateFields)
        .def("RenderGBuffer", [](LightingRenderer& r, const std::vector<float>& pos, const std::vector<float>& scale) { r.RenderGBuffer(v2(pos), v2(scale)); },
             py::arg("viewportPosition") = std::vector<float>{0, 0}, py::arg("viewportScale") = std::vector<float>{1, 1})
        .def("ReadGBuffer", [](const LightingRenderer& r) {
            py::array_t<float> out({ (py::ssize_t)r.Configuration.RenderHeight, (py::ssize_t)r.Configuration.RenderWidth, (py::ssize_t)4 });
            if (!r.Configuration.HighQualityGBuffer) throw ArgumentException("ReadGBuffer reads the Vector4 format");
            ThrowIfFailed(ilm_gbuffer_download(r.GBuffer(), out.mutable_data()));
            return out; })
        .def("BenchResolve", [](LightingRenderer& r, const std::string& hdrBytes, int format, int iterations) {
            // `iterations` resolves into a target of `format`, timed with the context's HIP events; returns ms per resolve
            if (hdrBytes.size() != sizeof(IlmHDRConfiguration)) throw ArgumentException("hdr must be an IlmHDRConfiguration");
            IlmHDRConfiguration hdr; std::memcpy(&hdr, hdrBytes.data(), sizeof(hdr));
            IlmHandle target = 0;
            ThrowIfFailed(ilm_lightmap_create(r.Context.Handle(), r.Configuration.RenderWidth, r.Configuration.RenderHeight, format, nullptr, &target));
            float ms = 0;
            try {
                r.Resolve(target, &hdr);
                r.Context.Sync();
                r.Context.TimerStart();
                for (int i = 0; i < iterations; i++) r.Resolve(target, &hdr);
                ms = r.Context.TimerStop() / (float)std::max(iterations, 1);
            } catch (...) { ilm_lightmap_destroy(target); throw; }
            ilm_lightmap_destroy(target);
            return ms;
        })
        .def("ResolveToArray", [](const LightingRenderer& r, py::object hdrBytes) {
            // Resolve into a float4 target owned by the renderer's context, read it back (test / bench plumbing)
            const int w = r.Configuration.RenderWidth, h = r.Configuration.RenderHeight;
            IlmHandle target = 0;
            ThrowIfFailed(ilm_lightmap_create(r.Context.Handle(), w, h, ILM_LIGHTMAP_FLOAT4, nullptr, &target));
            IlmHDRConfiguration hdr; const IlmHDRConfiguration* hp = nullptr;
            if (!hdrBytes.is_none()) {
                std::string b = hdrBytes.cast<std::string>();
                if (b.size() != sizeof(IlmHDRConfiguration)) { ilm_lightmap_destroy(target); throw ArgumentException("hdr must be an IlmHDRConfiguration"); }
                std::memcpy(&hdr, b.data(), sizeof(hdr)); hp = &hdr;
            }
            py::array_t<float> out({ (py::ssize_t)h, (py::ssize_t)w, (py::ssize_t)4 });
            try {
                r.Resolve(target, hp);
                ThrowIfFailed(ilm_lightmap_download(target, out.mutable_data(), 0, h));
            } catch (...) { ilm_lightmap_destroy(target); throw; }
            ilm_lightmap_destroy(target);
            return out;
        }, py::arg("hdr") = py::none())
        .def("Resolve", [](const LightingRenderer& r, IlmHandle destination, py::object hdrBytes, int rowBegin, int rowEnd) {
            if (hdrBytes.is_none()) { r.Resolve(destination, nullptr, rowBegin, rowEnd); return; }
            std::string b = hdrBytes.cast<std::string>();
            if (b.size() != sizeof(IlmHDRConfiguration)) throw ArgumentException("hdr must be an IlmHDRConfiguration");
            IlmHDRConfiguration h; std::memcpy(&h, b.data(), sizeof(h));
            r.Resolve(destination, &h, rowBegin, rowEnd);
        }, py::arg("destination"), py::arg("hdr") = py::none(), py::arg("rowBegin") = 0, py::arg("rowEnd") = -1)
        .def_property_readonly("Probes", [](LightingRenderer& r) -> LightProbeCollection& { return r.Probes; }, py::return_value_policy::reference_internal)
        .def_static("PackParticleLightBytes", [](const ParticleLightSource& pls, bool haveDF) {
            auto p = LightingRenderer::PackParticleLight(pls, haveDF); return py::bytes((const char*)&p, sizeof(p)); })
        .def("InvalidateFields", &LightingRenderer::InvalidateFields, py::arg("invalidateDistanceField") = true)
        .def("RenderLighting", [](LightingRenderer& r, float intensityScale, int rowBegin, int rowEnd, bool wantStats) -> py::object {
            if (!wantStats) { r.RenderLighting(intensityScale, rowBegin, rowEnd, nullptr); return py::none(); }
            IlmRenderStats st{};
            r.RenderLighting(intensityScale, rowBegin, rowEnd, &st);
            return py::make_tuple(st.SdfSamples, st.PixelLightPairs, st.TracedPairs);
        }, py::arg("intensityScale") = 1.0f, py::arg("rowBegin") = 0, py::arg("rowEnd") = -1, py::arg("wantStats") = false)
        .def("RenderLightingMany", [](LightingRenderer& r, int count, float intensityScale, int rowBegin, int rowEnd) {
            for (int i = 0; i < count; i++) r.RenderLighting(intensityScale, rowBegin, rowEnd, nullptr);
        })
        .def("ReadLightmap", [](const LightingRenderer& r, int firstRow, int rowCount) -> py::array {
            const int w = r.Configuration.RenderWidth;
            if (rowCount < 0) rowCount = r.Configuration.RenderHeight - firstRow;
            if (r.LightmapFormat() == ILM_LIGHTMAP_FLOAT4) {
                py::array_t<float> out({ (py::ssize_t)rowCount, (py::ssize_t)w, (py::ssize_t)4 });
                r.ReadLightmap(out.mutable_data(), firstRow, rowCount);
                return std::move(out);
            } else if (r.LightmapFormat() == ILM_LIGHTMAP_HALF4) {
                py::array_t<uint16_t> out({ (py::ssize_t)rowCount, (py::ssize_t)w, (py::ssize_t)4 });
                r.ReadLightmap(out.mutable_data(), firstRow, rowCount);
                return std::move(out);
            }
            py::array_t<uint8_t> out({ (py::ssize_t)rowCount, (py::ssize_t)w, (py::ssize_t)4 });
            r.ReadLightmap(out.mutable_data(), firstRow, rowCount);
            return std::move(out);
        }, py::arg("firstRow") = 0, py::arg("rowCount") = -1)
        .def_property_readonly("LightmapHandle", &LightingRenderer::Lightmap)
        .def_property_readonly("LightmapFormat", &LightingRenderer::LightmapFormat)
        .def("GetDistanceFieldUniformsBytes", [](const LightingRenderer& r) {
            auto u = r.GetDistanceFieldUniforms(r.Configuration.DefaultQuality); return py::bytes((const char*)&u, sizeof(u)); })
        .def("GetPackedLightVertices", [](const LightingRenderer& r) {
            const auto& v = r.PackedLightVertices(); return py::bytes((const char*)v.data(), v.size() * sizeof(IlmLightVertex)); })
        .def_property_readonly("LastLightCount", [](const LightingRenderer& r) { return r.PackedLightVertices().size(); })
        .def("GetEnvironmentUniformsBytes", [](const LightingRenderer& r) { auto u = r.GetEnvironmentUniforms(); return py::bytes((const char*)&u, sizeof(u)); })
        .def_static("PackSphereLightBytes", [](const SphereLightSource& l, float intensityScale, bool haveDF) -> py::object {
            IlmLightVertex v;
            if (!LightingRenderer::PackSphereLight(l, intensityScale, haveDF, v)) return py::none();
            return py::bytes((const char*)&v, sizeof(v));
        });
}

#!/usr/bin/env python3
"""Generate integration/IlluminantHip.cs -- the C# P/Invoke layer of include/illuminant_hip.h -- from the header itself.

    python tools/gen_csharp_binding.py            writes integration/IlluminantHip.cs
    python tools/gen_csharp_binding.py --check    exits 1 when the committed file is not what the header generates

The reference's toolchain (.NET 4.8 + XNA/FNA + Fracture) is not in this image, so the file cannot be compiled here; generating it
mechanically keeps every struct layout and every entry point in step with the header (tests/test_abi_layout.py runs --check, and
checks the header against the library and the ctypes mirrors).  The layout rules are C#'s own: LayoutKind.Sequential with Pack = 4
reproduces the C layout of these structs (only 4- and 8-byte scalars; every 8-byte member sits at an 8-byte offset by construction,
which the generator verifies), unions become LayoutKind.Explicit, fixed-size arrays become `fixed` buffers of their scalar type.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "illuminant_hip.h")
OUT = os.path.join(ROOT, "integration", "IlluminantHip.cs")

SCALARS = {"float": ("float", 4), "int32_t": ("int", 4), "uint32_t": ("uint", 4), "uint8_t": ("byte", 1), "uint64_t": ("ulong", 8),
           "int64_t": ("long", 8), "IlmHandle": ("ulong", 8), "uint16_t": ("ushort", 2), "double": ("double", 8)}
# structs the reference already has: used as they are (same byte layout)
REFERENCE_TYPES = {"IlmFloat4": ("Vector4", 16), "IlmMatrix": ("Matrix", 64), "IlmLightVertex": ("LightVertex", 128)}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def parse_structs(text):
    """[(name, kind, [(ctype, field, dims)])] in declaration order; kind = 'struct'.  Anonymous unions become a nested entry."""
    structs = []
    for m in re.finditer(r"typedef struct (Ilm\w+) \{(.*?)\} \1;", text, flags=re.S):
        name, body = m.group(1), m.group(2)
        fields = []
        union = re.search(r"union \{(.*?)\} (\w+);", body, flags=re.S)
        if union:
            ufields = parse_fields(union.group(1))
            body = body.replace(union.group(0), "@UNION %s;" % union.group(2))
        for ctype, fname, dims in parse_fields(body):
            if ctype == "@UNION":
                fields.append(("@UNION", fname, ufields))
            else:
                fields.append((ctype, fname, dims))
        structs.append((name, fields))
    return structs


def parse_fields(body):
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(@UNION|[A-Za-z_]\w*)\s+(.*)$", decl)
        ctype, rest = m.group(1), m.group(2)
        for item in rest.split(","):
            item = item.strip()
            dims = [int(d) if d.isdigit() else d for d in re.findall(r"\[(\w+)\]", item)]
            fname = re.match(r"\w+", item).group(0)
            out.append((ctype, fname, dims))
    return out


def parse_defines(text):
    return {k: int(v) for k, v in re.findall(r"#define (ILM_\w+)\s+\(?(-?\d+)u?\)?\s*$", text, flags=re.M)}


def parse_float_defines(text):
    return {k: v for k, v in re.findall(r"#define (ILM_\w+)\s+(-?\d+\.\d+f)\s*$", text, flags=re.M)}


def parse_enums(text):
    out = []
    for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", text, flags=re.S):
        value = -1
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = [x.strip() for x in item.split("=")]
                value = int(v, 0)
            else:
                k, value = item, value + 1
            out.append((k, value))
    return out


def parse_functions(text):
    fns = []
    for m in re.finditer(r"^\s*(const char\*|int32_t)\s+(ilm_\w+)\s*\((.*?)\)\s*;", text, flags=re.S | re.M):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        fns.append((ret, name, [] if args in ("", "void") else [a.strip() for a in args.split(",")]))
    return fns


class Layout:
    def __init__(self, structs, defines):
        self.defines = defines
        self.sizes = {k: v[1] for k, v in SCALARS.items()}
        self.sizes.update({k: v[1] for k, v in REFERENCE_TYPES.items()})
        self.aligns = {k: min(v[1], 8) for k, v in SCALARS.items()}
        self.aligns.update({k: 4 for k in REFERENCE_TYPES})
        self.structs = dict(structs)
        for name, fields in structs:
            self.size_of_struct(name, fields)

    def dim(self, d):
        return d if isinstance(d, int) else self.defines[d]

    def count(self, dims):
        n = 1
        for d in dims:
            n *= self.dim(d)
        return n

    def size_of_struct(self, name, fields):
        if name in self.sizes:
            return self.sizes[name]
        off, align = 0, 1
        for ctype, fname, dims in fields:
            if ctype == "@UNION":
                usize = max(self.sizes[t] * self.count(d) for t, _, d in dims)
                ualign = max(self.aligns[t] for t, _, d in dims)
                off = (off + ualign - 1) // ualign * ualign
                off += usize
                align = max(align, ualign)
                continue
            a, s = self.aligns[ctype], self.sizes[ctype]
            assert off % a == 0, "%s.%s would need padding: Pack = 4 sequential layout would differ from C" % (name, fname)
            off += s * self.count(dims)
            align = max(align, a)
        assert off % align == 0, "%s has tail padding" % name
        self.sizes[name], self.aligns[name] = off, align
        return off


def cs_type(ctype):
    if ctype in SCALARS:
        return SCALARS[ctype][0]
    if ctype in REFERENCE_TYPES:
        return REFERENCE_TYPES[ctype][0]
    return ctype


def scalar_of(ctype, layout):
    """(C# scalar type, scalars per element) for a fixed buffer of `ctype` elements."""
    if ctype in SCALARS:
        return SCALARS[ctype][0], 1
    return "float", layout.sizes[ctype] // 4      # IlmFloat4 / IlmMatrix arrays: fixed float buffers


def emit_struct(name, fields, layout, out):
    unsafe = any((ctype != "@UNION" and dims) for ctype, _, dims in fields)
    has_union = any(ctype == "@UNION" for ctype, _, _ in fields)
    if has_union:
        # header of the union's owner: sequential members first, then the overlapping ones at one explicit offset
        out.append("    [StructLayout(LayoutKind.Explicit, Size = %d)]" % layout.sizes[name])
        out.append("    public %sstruct %s {" % ("unsafe " if unsafe else "", name))
        off = 0
        for ctype, fname, dims in fields:
            if ctype == "@UNION":
                for utype, ufield, _ in dims:
                    out.append("        [FieldOffset(%d)] public %s %s;" % (off, cs_type(utype), ufield))
                off += max(layout.sizes[t] for t, _, _ in dims)
            elif dims:
                sc, per = scalar_of(ctype, layout)
                out.append("        [FieldOffset(%d)] public fixed %s %s[%d];" % (off, sc, fname, layout.count(dims) * per))
                off += layout.sizes[ctype] * layout.count(dims)
            else:
                out.append("        [FieldOffset(%d)] public %s %s;" % (off, cs_type(ctype), fname))
                off += layout.sizes[ctype]
        out.append("    }")
        return
    out.append("    [StructLayout(LayoutKind.Sequential, Pack = 4, Size = %d)]" % layout.sizes[name])
    out.append("    public %sstruct %s {" % ("unsafe " if unsafe else "", name))
    for ctype, fname, dims in fields:
        if not dims:
            out.append("        public %s %s;" % (cs_type(ctype), fname))
        elif ctype in SCALARS or ctype in REFERENCE_TYPES:
            sc, per = scalar_of(ctype, layout)
            note = "" if per == 1 else "   // %s[%s]" % (cs_type(ctype), "][".join(str(layout.dim(d)) for d in dims))
            out.append("        public fixed %s %s[%d];%s" % (sc, fname, layout.count(dims) * per, note))
        else:
            # array of a generated struct: C# has no fixed buffers of structs -- spell the elements out
            for i in range(layout.count(dims)):
                out.append("        public %s %s%d;" % (cs_type(ctype), fname, i))
    out.append("    }")


def cs_arg(arg):
    arg = arg.replace("const ", "")
    m = re.match(r"(.+?)\s*(\**)\s*(\w+)(\[\d*\])?$", arg)
    ctype, stars, name, arr = m.group(1).strip(), m.group(2), m.group(3), m.group(4)
    if arr:
        stars += "*"
    base = "void" if ctype == "void" else ("byte" if ctype == "char" else cs_type(ctype))
    cname = re.sub(r"_(\w)", lambda k: k.group(1).upper(), name)
    if cname in ("out", "in", "ref", "params", "object", "lock", "base", "event"):
        cname = "@" + cname
    return "%s%s %s" % (base, stars, cname)


def generate():
    raw = open(HEADER).read()
    text = strip_comments(raw)
    defines = parse_defines(text)
    structs = parse_structs(text)
    layout = Layout(structs, defines)
    fns = parse_functions(text)
    out = []
    out.append("// IlluminantHip.cs -- P/Invoke layer of libilluminant_hip.so for sq/Illuminant (drop into Illuminant/Native/).")
    out.append("// GENERATED from include/illuminant_hip.h by tools/gen_csharp_binding.py -- do not edit; the header carries the documentation")
    out.append("// and the reference file:line each entry point replaces.  ABI version %d." % defines["ILM_ABI_VERSION"])
    out.append("//")
    out.append("// Vector4 / Matrix are XNA's; LightVertex is Illuminant/Vertices.cs:10-39; the Uniforms.* structs of the reference")
    out.append("// (Uniforms.cs:14-24,79-88,197-236; Bezier.cs:433-441,588-599) have the byte layout of the Ilm* mirrors below and can be passed")
    out.append("// with a pointer cast.  Every call returns 0 or an error code: IlluminantHip.Check turns it into the reference's exception types.")
    out.append("using System;")
    out.append("using System.Runtime.InteropServices;")
    out.append("using Microsoft.Xna.Framework;")
    out.append("")
    out.append("namespace Squared.Illuminant.Native {")
    out.append("    public sealed class IlluminantHipException : Exception {")
    out.append("        public readonly int Code;")
    out.append("        public IlluminantHipException (int code, string message) : base(message) { Code = code; }")
    out.append("    }")
    out.append("")
    out.append("    public static class IlmConstants {")
    for k in sorted(defines):
        out.append("        public const int %s = %d;" % (k[4:], defines[k]))
    for k, v in sorted(parse_float_defines(text).items()):
        out.append("        public const float %s = %s;" % (k[4:], v))
    for k, v in parse_enums(text):
        out.append("        public const int %s = %d;" % (k[4:], v))
    out.append("    }")
    out.append("")
    for name, fields in structs:
        if name in REFERENCE_TYPES:
            continue
        emit_struct(name, fields, layout, out)
        out.append("")
    out.append("    internal static unsafe class IlluminantHip {")
    out.append('        const string Lib = "illuminant_hip";             // libilluminant_hip.so next to the game\'s assemblies')
    out.append("")
    for ret, name, args in fns:
        cs_ret = "IntPtr" if ret.startswith("const char") else "int"
        out.append("        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern %s %s (%s);"
                   % (cs_ret, name, ", ".join(cs_arg(a) for a in args)))
    out.append("")
    out.append("        public static void Check (int code) {")
    out.append("            if (code == 0) return;")
    out.append("            var msg = Marshal.PtrToStringAnsi(ilm_last_error());")
    out.append("            // the reference's own exception types for the conditions it checks itself")
    out.append("            if (code == IlmConstants.ERR_TOO_MANY) throw new InvalidOperationException(msg);   // \"Maximum number of attractors per instance is 16\" (Transforms.cs:348-349)")
    out.append("            if (code == IlmConstants.ERR_STATE) throw new InvalidOperationException(msg);      // distance field update without a field (ParticleSystem.cs:835-836)")
    out.append("            throw new IlluminantHipException(code, msg);")
    out.append("        }")
    out.append("    }")
    out.append("}")
    return "\n".join(out) + "\n", layout, fns, structs


def main():
    text, _, _, _ = generate()
    if "--check" in sys.argv:
        if not os.path.exists(OUT) or open(OUT).read() != text:
            print("integration/IlluminantHip.cs is out of date: run python tools/gen_csharp_binding.py")
            return 1
        return 0
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote %s (%d lines)" % (OUT, text.count("\n")))
    return 0


if __name__ == "__main__":
    sys.exit(main())

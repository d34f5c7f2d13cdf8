"""The Julia host module cannot be executed in the build image (no Julia toolchain), so its binding table is checked mechanically against
include/rdhip.h: every exported `rdhip_*` entry point must have a `ccall((:name, librdhip), ...)` in RobotDynamicsHIP.jl whose return type and
argument-type tuple are compatible with the C prototype, argument for argument; every compute entry point must be reachable with device
arrays (memspace argument passed as a variable, not the HOST constant); and the host-side surface the reference's callers use (model types,
RK2, v0.3 aliases) must be present."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rdhip.h")
JL = os.path.join(ROOT, "robotdynamics.jl_amd", "julia", "RobotDynamicsHIP.jl")


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def c_prototypes():
    txt = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|void\*|const char\*)\s+(rdhip_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        protos[name] = (ret, [] if args in ("void", "") else split_top(args))
    return protos


def jl_ccalls():
    txt = open(JL).read()
    calls = {}
    for m in re.finditer(r"ccall\(\(:(rdhip_[a-z0-9_]+), librdhip\),\s*([A-Za-z{}]+),\s*\(", txt):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(txt[i], 0)
            i += 1
        calls.setdefault(m.group(1), []).append((m.group(2), split_top(txt[m.end():i - 1]), txt[i:i + 900]))
    return calls


def c_kind(arg):
    """coarse class of a C parameter"""
    a = arg.replace("const ", "").strip()
    if "*" in a or "[" in a:
        if a.startswith("char") and "[" not in a:
            return "cstring"
        return "ptr"
    base = a.rsplit(" ", 1)[0] if " " in a else a
    return {"int": "int", "long long": "longlong", "unsigned long long": "ulonglong", "double": "double", "float": "float"}[base.strip()]


JL_OK = {"int": {"Cint"}, "longlong": {"Clonglong"}, "ulonglong": {"Culonglong"}, "double": {"Cdouble"}, "float": {"Cfloat"}, "cstring": {"Cstring", "Ptr{UInt8}"}}


def test_every_entry_point_has_a_matching_ccall():
    protos, calls = c_prototypes(), jl_ccalls()
    assert len(protos) >= 50
    missing = sorted(set(protos) - set(calls))
    assert not missing, f"no ccall in RobotDynamicsHIP.jl for: {missing}"
    unknown = sorted(set(calls) - set(protos))
    assert not unknown, f"ccall to symbols the header does not declare: {unknown}"
    for name, (ret, cargs) in protos.items():
        for jret, jargs, _ in calls[name]:
            assert jret == {"int": "Cint", "void*": "Ptr{Cvoid}", "const char*": "Cstring"}[ret], f"{name}: return type {jret}"
            assert len(jargs) == len(cargs), f"{name}: header has {len(cargs)} parameters, the ccall passes {len(jargs)}"
            for k, (ca, ja) in enumerate(zip(cargs, jargs)):
                kind = c_kind(ca)
                if kind == "ptr":
                    assert ja.startswith(("Ptr{", "Ref{")), f"{name} argument {k + 1} ({ca}): {ja} is not a pointer type"
                else:
                    assert ja in JL_OK[kind], f"{name} argument {k + 1} ({ca}): {ja} does not match"


def test_every_batched_operator_has_a_device_path():
    """The round-1 module passed RDHIP_HOST everywhere (a 136 MB device-to-host copy per batch).  Now the memspace argument of every compute
    entry point must be computed from the arrays (`sp`), i.e. DeviceArray arguments run in place on the GPU."""
    calls = jl_ccalls()
    compute = ["rdhip_dynamics", "rdhip_discrete_dynamics", "rdhip_jacobian", "rdhip_jacobian_multi", "rdhip_dynamics_error", "rdhip_dynamics_error_jacobian",
               "rdhip_error_expansion", "rdhip_errstate_jacobian", "rdhip_state_diff", "rdhip_state_diff_map", "rdhip_nabla_errstate_jacobian", "rdhip_rollout", "rdhip_rollout_layout", "rdhip_rollout_split", "rdhip_rollout_jacobian",
               "rdhip_rollout_jacobian_multi"]
    for name in compute:
        for _, _, tail in calls[name]:
            call_args = tail.split("check(")[0]
            assert re.search(r",\s*sp\)", call_args), f"{name}: memspace is not derived from the arguments"
            assert "HOST)" not in call_args, f"{name}: hard-wired RDHIP_HOST"
    txt = open(JL).read()
    for needle in ("errmapcode(::Rotations.CayleyMap)", "errmapcode(::Rotations.ExponentialMap)", "errmapcode(::Rotations.MRPMap)", "errmapcode(::Rotations.QuatVecMap)",
                   "errmapcode(::Rotations.IdentityMap)", "errmap::Rotations.ErrorMap=Rotations.CayleyMap()"):      # state_diff's optional error map (src/liestate.jl:26-32)
        assert needle in txt, needle
    for needle in ("mutable struct DeviceArray", "struct DeviceTrajectory", "rdhip_malloc", "rdhip_memcpy_h2d", "rdhip_memcpy_d2h", "rdhip_graph_launch"):
        assert needle in txt


def test_host_surface_of_the_reference_is_present():
    txt = open(JL).read()
    for model in ("Cartpole", "DoubleIntegrator", "LinearModel", "Satellite", "Quadrotor", "SatellitePair", "LieSatellite", "CustomModel", "CustomLieModel"):
        assert re.search(rf"struct {model}\b", txt), model
    assert "struct RK2 <: RobotDynamics.Explicit" in txt and "RobotDynamics.integrate(::RK2" in txt and "RobotDynamics.jacobian!(::RK2" in txt
    for fn in ("RobotDynamics.jacobian!", "RobotDynamics.evaluate!", "RobotDynamics.discrete_dynamics!", "RobotDynamics.dynamics!", "RobotDynamics.errstate_jacobian!",
               "RobotDynamics.state_diff!", "RobotDynamics.∇errstate_jacobian!", "RobotDynamics.rollout!", "RobotDynamics.dynamics_error_jacobian!",
               "RobotDynamics.dynamics_error(", "error_expansion!", "rollout_jacobian!", "jacobian_multi!", "jacobian_sharded!", "gather!",
               "discrete_jacobian!", "state_diff_jacobian!", "∇²differential!", "traj_size"):
        assert fn in txt, fn
    # the model enums agree with the header
    hdr = open(HEADER).read()
    for k, name in enumerate(["CARTPOLE", "DOUBLE_INTEGRATOR", "LINEAR", "SATELLITE", "QUADROTOR", "SATELLITE_PAIR", "LIE_SATELLITE", "CUSTOM"]):
        assert re.search(rf"RDHIP_MODEL_{name} = {k}\b", hdr)
    assert re.search(r"const MODEL_CARTPOLE, MODEL_DOUBLE_INTEGRATOR, MODEL_LINEAR, MODEL_SATELLITE, MODEL_QUADROTOR, MODEL_SATELLITE_PAIR, MODEL_LIE_SATELLITE, MODEL_CUSTOM = 0:7", txt)


def test_brackets_balance():
    """cheap syntax sanity of a file nobody can run here: (), [], {} balance outside strings / comments, and every block opener has its `end`"""
    txt = open(JL).read()
    txt = re.sub(r'"""(.*?)"""', '""', txt, flags=re.S)
    code = []
    for line in txt.split("\n"):
        line = re.sub(r'"(\\.|[^"\\])*"', '""', line)
        code.append(line.split("#")[0])
    src = "\n".join(code)
    for o, c in ("()", "[]", "{}"):
        assert src.count(o) == src.count(c), f"unbalanced {o}{c}: {src.count(o)} vs {src.count(c)}"
    openers = 0
    for line in src.split("\n"):
        st = line.strip()
        if re.match(r"(?:RobotDynamics\.@autodiff\s+)?(?:mutable\s+)?struct\b|abstract type\b|(?:function|if|for|while|try|let|module|quote|macro)\b", st):
            openers += 1
        openers += len(re.findall(r"\bbegin\b", st)) + (1 if re.search(r"\bdo\b(\s+\w+)?\s*$", st) else 0)
    ends = len(re.findall(r"(?<![\w.:\[])end\b", src))
    assert openers == ends, f"{openers} block openers vs {ends} `end`s"

"""MJCF subset compiler: rodent.xml (+ env edits) -> numeric model constants.

Host-side, runs once per model.  Replaces, for the hot path only, the chain the
reference runs at construction time (reference envs/rodent.py:39-63):

    dm_control.mjcf.from_path -> actuator rewrite (rodent.py:41-45)
      -> rescale.rescale_subtree(root, 0.9, 0.9) (rodent.py:47-52)
      -> MuJoCo compile (rodent.py:53) -> option edits (rodent.py:55-63)

None of dm_control / mujoco is available here, so the subset of MJCF that the
reference's assets use is restated:

  * nested <default class=...> inheritance (geom / joint / general), <freejoint>
    takes no defaults;
  * orientation via quat= or euler= (radians, intrinsic xyz);
  * primitive geoms sphere / capsule / ellipsoid / box / plane, mass and inertia
    from density, body inertial frame by merging its geoms;
  * hinge and free joints, joint limits, stiffness / damping / armature /
    springref, solreflimit / solimplimit;
  * <general> actuators with joint transmission and filter dynamics;
  * constants MuJoCo derives at qpos0: dof_invweight0, body_invweight0,
    stat.meaninertia.

rescale semantics [dm_control rescale_subtree]: only attributes WRITTEN on an
element inside worldbody/body are scaled (pos, size); values inherited from a
default class are untouched.  This reading reproduces the FK golden of the
shipped clip to ~1e-8 (tests/test_model_golden.py); scaling inherited values
does not.

All arithmetic here is float64; consumers cast to float32.
"""
from __future__ import annotations

import dataclasses
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional, Sequence

import numpy as np

MJ_MINVAL = 1e-15

GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_BOX = 0, 2, 3, 4, 6
_GEOM_TYPES = {
    "plane": GEOM_PLANE,
    "sphere": GEOM_SPHERE,
    "capsule": GEOM_CAPSULE,
    "ellipsoid": GEOM_ELLIPSOID,
    "box": GEOM_BOX,
}
JNT_FREE, JNT_HINGE = 0, 3


# ----------------------------------------------------------------------------
# small quaternion helpers (w, x, y, z)
# ----------------------------------------------------------------------------
def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ]
    )


def quat_to_mat(q):
    w, x, y, z = q
    return np.array(
        [
            [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
        ]
    )


def mat_to_quat(m):
    """Rotation matrix -> unit quaternion (w >= 0 branch-stable)."""
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = np.array([(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s])
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = np.array([(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s])
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = np.array([(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s])
    return q / np.linalg.norm(q)


def axis_angle_quat(axis, angle):
    s, c = np.sin(0.5 * angle), np.cos(0.5 * angle)
    return np.array([c, axis[0] * s, axis[1] * s, axis[2] * s])


def euler_to_quat(e):
    """MuJoCo default eulerseq 'xyz' (lower case = intrinsic): q = qx * qy * qz."""
    q = np.array([1.0, 0.0, 0.0, 0.0])
    for i in range(3):
        ax = np.zeros(3)
        ax[i] = 1.0
        q = quat_mul(q, axis_angle_quat(ax, e[i]))
    return q


def rotate(v, q):
    return quat_to_mat(q) @ v


# ----------------------------------------------------------------------------
# XML + defaults
# ----------------------------------------------------------------------------
def _floats(s: Optional[str]) -> Optional[np.ndarray]:
    if s is None:
        return None
    return np.array([float(x) for x in s.split()], dtype=np.float64)


class _Defaults:
    """Resolved default classes: class name -> {tag: {attr: str}}."""

    def __init__(self, root: ET.Element):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {"main": {}}
        top = root.find("default")
        if top is not None:
            self._walk(top, "main", {})

    def _walk(self, node: ET.Element, name: str, inherited: Dict[str, Dict[str, str]]):
        cur = {tag: dict(attrs) for tag, attrs in inherited.items()}
        for child in node:
            if child.tag == "default":
                continue
            cur.setdefault(child.tag, {}).update(child.attrib)
        self.classes[name] = cur
        for child in node:
            if child.tag == "default":
                self._walk(child, child.attrib["class"], cur)

    def resolve(self, elem: ET.Element, tag: Optional[str] = None, childclass: Optional[str] = None) -> Dict[str, str]:
        """Attributes of `elem` after default-class resolution: its own `class`, else the `childclass` of the nearest
        enclosing body that sets one, else the top-level defaults."""
        tag = tag or elem.tag
        cls = elem.attrib.get("class", childclass or "main")
        out = dict(self.classes.get(cls, {}).get(tag, {}))
        out.update(elem.attrib)
        return out


# ----------------------------------------------------------------------------
# primitive-geom mass / inertia  [MuJoCo user_objects: mjCGeom::GetVolume/SetInertia]
# ----------------------------------------------------------------------------
def _geom_volume(gtype: int, size: np.ndarray) -> float:
    if gtype == GEOM_SPHERE:
        return 4.0 / 3.0 * np.pi * size[0] ** 3
    if gtype == GEOM_CAPSULE:
        return np.pi * size[0] ** 2 * (2 * size[1]) + 4.0 / 3.0 * np.pi * size[0] ** 3
    if gtype == GEOM_ELLIPSOID:
        return 4.0 / 3.0 * np.pi * size[0] * size[1] * size[2]
    if gtype == GEOM_BOX:
        return 8.0 * size[0] * size[1] * size[2]
    return 0.0


def _geom_inertia(gtype: int, size: np.ndarray, mass: float) -> np.ndarray:
    """Diagonal inertia in the geom frame."""
    if gtype == GEOM_SPHERE:
        v = 2.0 * mass * size[0] ** 2 / 5.0
        return np.array([v, v, v])
    if gtype == GEOM_CAPSULE:
        r, h = size[0], 2 * size[1]
        vol = _geom_volume(gtype, size)
        sphere_mass = mass * (4.0 / 3.0 * np.pi * r**3) / vol
        cyl_mass = mass - sphere_mass
        ixy = cyl_mass * (3 * r * r + h * h) / 12.0
        iz = cyl_mass * r * r / 2.0
        sphere_i = 2.0 * sphere_mass * r * r / 5.0
        ixy += sphere_i + sphere_mass * h * (3 * r + 2 * h) / 8.0
        iz += sphere_i
        return np.array([ixy, ixy, iz])
    if gtype == GEOM_ELLIPSOID:
        a, b, c = size[:3]
        return mass / 5.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == GEOM_BOX:
        a, b, c = size[:3]
        return mass / 3.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    return np.zeros(3)


@dataclasses.dataclass
class _Geom:
    name: str
    body: int
    gtype: int
    size: np.ndarray
    pos: np.ndarray
    quat: np.ndarray
    density: float
    contype: int
    conaffinity: int
    condim: int
    priority: int
    friction: np.ndarray
    solref: np.ndarray
    solimp: np.ndarray
    solmix: float
    margin: float
    gap: float
    mass: float = 0.0


def _full_solimp(v: Optional[np.ndarray]) -> np.ndarray:
    out = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
    if v is not None:
        out[: len(v)] = v
    return out


@dataclasses.dataclass
class CompiledModel:
    """Numeric constants; field names follow MuJoCo's mjModel where one exists."""

    names: Dict[str, List[str]]
    arrays: Dict[str, np.ndarray]
    scalars: Dict[str, float]

    def __getattr__(self, k):
        d = object.__getattribute__(self, "__dict__")
        if k in d.get("arrays", {}):
            return d["arrays"][k]
        if k in d.get("scalars", {}):
            return d["scalars"][k]
        raise AttributeError(k)

    def body_id(self, name: str) -> int:
        return self.names["body"].index(name)

    def joint_id(self, name: str) -> int:
        return self.names["joint"].index(name)

    # ---- (de)serialisation: plain npz, data only -------------------------
    def save(self, path: str) -> None:
        payload = {f"a_{k}": v for k, v in self.arrays.items()}
        payload.update({f"s_{k}": np.float64(v) for k, v in self.scalars.items()})
        payload.update({f"n_{k}": np.array(v) for k, v in self.names.items()})
        np.savez_compressed(path, **payload)

    @staticmethod
    def load(path: str) -> "CompiledModel":
        z = np.load(path, allow_pickle=False)
        arrays, scalars, names = {}, {}, {}
        for k in z.files:
            if k.startswith("a_"):
                arrays[k[2:]] = z[k]
            elif k.startswith("s_"):
                scalars[k[2:]] = float(z[k])
            elif k.startswith("n_"):
                names[k[2:]] = [str(x) for x in z[k]]
        return CompiledModel(names=names, arrays=arrays, scalars=scalars)


def compile_mjcf(
    xml_path: str,
    scale_factor: Optional[float] = 0.9,
    torque_actuators: bool = True,
    iterations: int = 6,
    ls_iterations: int = 6,
    solver: str = "cg",
    rescale_defaults: bool = False,
) -> CompiledModel:
    """Compile an MJCF file the way reference envs/rodent.py:39-63 does.

    scale_factor=None skips the dm_control rescale.  `rescale_defaults=True`
    is a diagnostic switch (scales default-class pos too) used only by the
    golden test to show the explicit-only reading is the discriminating one.
    """
    root = ET.parse(xml_path).getroot()
    defaults = _Defaults(root)
    comp = root.find("compiler")
    angle_rad = comp is not None and comp.attrib.get("angle", "degree") == "radian"
    ang = 1.0 if angle_rad else np.pi / 180.0
    sf = 1.0 if scale_factor is None else float(scale_factor)

    body_names: List[str] = ["world"]
    body_parent = [0]
    body_pos = [np.zeros(3)]
    body_quat = [np.array([1.0, 0, 0, 0])]
    geoms: List[_Geom] = []
    joints = []  # dicts

    def orient(attrs) -> np.ndarray:
        if "quat" in attrs:
            q = _floats(attrs["quat"])
            return q / np.linalg.norm(q)
        if "euler" in attrs:
            return euler_to_quat(_floats(attrs["euler"]) * ang)
        return np.array([1.0, 0, 0, 0])

    def scaled(elem: ET.Element, attrs: Dict[str, str], key: str) -> Optional[np.ndarray]:
        """Attribute value after the explicit-only rescale rule."""
        v = _floats(attrs.get(key))
        if v is None:
            return None
        if key in elem.attrib or rescale_defaults:
            v = v * sf
        return v

    def zaxis_quat(z: np.ndarray) -> np.ndarray:
        """quaternion rotating (0,0,1) onto z  [MuJoCo mjuu_z2quat]"""
        z = z / np.linalg.norm(z)
        axis = np.cross([0, 0, 1.0], z)
        s = np.linalg.norm(axis)
        if s < 1e-10:
            return np.array([1.0, 0, 0, 0]) if z[2] > 0 else np.array([0.0, 1.0, 0, 0])
        return axis_angle_quat(axis / s, np.arctan2(s, z[2]))

    def add_geom(elem: ET.Element, body: int, childclass: Optional[str] = None):
        a = defaults.resolve(elem, childclass=childclass)
        gtype = _GEOM_TYPES[a.get("type", "sphere")]
        size = scaled(elem, a, "size")
        size = np.zeros(3) if size is None else np.concatenate([size, np.zeros(3)])[:3]
        if "fromto" in a:
            ft = _floats(a["fromto"])
            if "fromto" in elem.attrib:
                mid, half = sf * 0.5 * (ft[3:] + ft[:3]), sf * 0.5 * (ft[3:] - ft[:3])
                ft = np.concatenate([mid - half, mid + half])
            vec = ft[3:] - ft[:3]
            length = np.linalg.norm(vec)
            pos = 0.5 * (ft[:3] + ft[3:])
            z = vec / length
            # quaternion rotating (0,0,1) onto z  [MuJoCo mjuu_z2quat]
            axis = np.cross([0, 0, 1.0], z)
            s = np.linalg.norm(axis)
            if s < 1e-10:
                quat = np.array([1.0, 0, 0, 0]) if z[2] > 0 else np.array([0.0, 1.0, 0, 0])
            else:
                quat = axis_angle_quat(axis / s, np.arctan2(s, z[2]))
            size = np.array([size[0], 0.5 * length, 0.0])
        else:
            pos = scaled(elem, a, "pos")
            pos = np.zeros(3) if pos is None else pos
            quat = zaxis_quat(_floats(a["zaxis"])) if "zaxis" in a else orient(a)
        fr = np.array([1.0, 0.005, 0.0001])
        f = _floats(a.get("friction"))
        if f is not None:
            fr[: len(f)] = f
        sr = np.array([0.02, 1.0])
        s_ = _floats(a.get("solref"))
        if s_ is not None:
            sr[: len(s_)] = s_
        g = _Geom(
            name=a.get("name", f"geom{len(geoms)}"),
            body=body,
            gtype=gtype,
            size=size,
            pos=pos,
            quat=quat,
            density=float(a.get("density", 1000.0)),
            contype=int(a.get("contype", 1)),
            conaffinity=int(a.get("conaffinity", 1)),
            condim=int(a.get("condim", 3)),
            priority=int(a.get("priority", 0)),
            friction=fr,
            solref=sr,
            solimp=_full_solimp(_floats(a.get("solimp"))),
            solmix=float(a.get("solmix", 1.0)),
            margin=float(a.get("margin", 0.0)),
            gap=float(a.get("gap", 0.0)),
        )
        if "mass" in a:
            g.mass = float(a["mass"])
        else:
            g.mass = g.density * _geom_volume(gtype, size)
        geoms.append(g)

    def add_joint(elem: ET.Element, body: int, childclass: Optional[str] = None):
        if elem.tag == "freejoint":
            joints.append(dict(name=elem.attrib.get("name", ""), type=JNT_FREE, body=body))
            return
        a = defaults.resolve(elem, "joint", childclass=childclass)
        jtype = a.get("type", "hinge")
        if jtype == "free":
            joints.append(dict(name=a.get("name", ""), type=JNT_FREE, body=body))
            return
        if jtype != "hinge":
            raise NotImplementedError(f"joint type {jtype}")
        pos = scaled(elem, a, "pos")
        axis = _floats(a.get("axis", "0 0 1"))
        rng = _floats(a.get("range"))
        limited = a.get("limited", "auto")
        is_limited = (limited == "true") or (limited == "auto" and rng is not None)
        sr = _floats(a.get("solreflimit", "0.02 1"))
        joints.append(
            dict(
                name=a.get("name", ""),
                type=JNT_HINGE,
                body=body,
                pos=np.zeros(3) if pos is None else pos,
                axis=axis / np.linalg.norm(axis),
                range=(rng * ang) if rng is not None else np.zeros(2),
                limited=bool(is_limited and rng is not None),
                stiffness=float(a.get("stiffness", 0.0)),
                damping=float(a.get("damping", 0.0)),
                armature=float(a.get("armature", 0.0)),
                springref=float(a.get("springref", 0.0)) * ang,
                ref=float(a.get("ref", 0.0)) * ang,
                margin=float(a.get("margin", 0.0)),
                solref=sr,
                solimp=_full_solimp(_floats(a.get("solimplimit"))),
            )
        )

    def walk(elem: ET.Element, body: int, childclass: Optional[str] = None):
        # MuJoCo numbering: a body's joints and geoms are registered when the body is
        # visited (document order within the body), then children depth-first.
        for child in elem:
            if child.tag in ("joint", "freejoint"):
                add_joint(child, body, childclass)
            elif child.tag == "geom":
                add_geom(child, body, childclass)
        for child in elem:
            if child.tag == "body":
                bid = len(body_names)
                body_names.append(child.attrib.get("name", f"body{bid}"))
                body_parent.append(body)
                p = _floats(child.attrib.get("pos"))
                body_pos.append(np.zeros(3) if p is None else p * sf)
                body_quat.append(orient(child.attrib))
                walk(child, bid, child.attrib.get("childclass", childclass))  # `childclass`: defaults of the subtree

    walk(root.find("worldbody"), 0)

    nbody = len(body_names)
    body_parent = np.array(body_parent, dtype=np.int32)
    body_pos = np.array(body_pos)
    body_quat = np.array(body_quat)

    # --- joints / dofs --------------------------------------------------------
    njnt = len(joints)
    jnt_type = np.array([j["type"] for j in joints], dtype=np.int32)
    jnt_bodyid = np.array([j["body"] for j in joints], dtype=np.int32)
    jnt_qposadr = np.zeros(njnt, dtype=np.int32)
    jnt_dofadr = np.zeros(njnt, dtype=np.int32)
    nq = nv = 0
    for i, j in enumerate(joints):
        jnt_qposadr[i], jnt_dofadr[i] = nq, nv
        if j["type"] == JNT_FREE:
            nq, nv = nq + 7, nv + 6
        else:
            nq, nv = nq + 1, nv + 1
    jnt_pos = np.zeros((njnt, 3))
    jnt_axis = np.zeros((njnt, 3))
    jnt_axis[:, 2] = 1.0
    jnt_range = np.zeros((njnt, 2))
    jnt_limited = np.zeros(njnt, dtype=np.int32)
    jnt_stiffness = np.zeros(njnt)
    jnt_margin = np.zeros(njnt)
    jnt_solref = np.tile(np.array([0.02, 1.0]), (njnt, 1))
    jnt_solimp = np.tile(_full_solimp(None), (njnt, 1))
    qpos0 = np.zeros(nq)
    qpos_spring = np.zeros(nq)
    dof_bodyid = np.zeros(nv, dtype=np.int32)
    dof_jntid = np.zeros(nv, dtype=np.int32)
    dof_armature = np.zeros(nv)
    dof_damping = np.zeros(nv)
    for i, j in enumerate(joints):
        qa, da = jnt_qposadr[i], jnt_dofadr[i]
        if j["type"] == JNT_FREE:
            qpos0[qa : qa + 3] = body_pos[j["body"]]
            qpos0[qa + 3 : qa + 7] = body_quat[j["body"]]
            qpos_spring[qa : qa + 7] = qpos0[qa : qa + 7]
            dof_bodyid[da : da + 6] = j["body"]
            dof_jntid[da : da + 6] = i
        else:
            jnt_pos[i], jnt_axis[i] = j["pos"], j["axis"]
            jnt_range[i], jnt_limited[i] = j["range"], int(j["limited"])
            jnt_stiffness[i], jnt_margin[i] = j["stiffness"], j["margin"]
            jnt_solref[i], jnt_solimp[i] = j["solref"], j["solimp"]
            qpos0[qa] = j["ref"]
            qpos_spring[qa] = j["springref"]
            dof_bodyid[da], dof_jntid[da] = j["body"], i
            dof_armature[da], dof_damping[da] = j["armature"], j["damping"]

    body_jntadr = np.full(nbody, -1, dtype=np.int32)
    body_jntnum = np.zeros(nbody, dtype=np.int32)
    body_dofadr = np.full(nbody, -1, dtype=np.int32)
    body_dofnum = np.zeros(nbody, dtype=np.int32)
    for i, j in enumerate(joints):
        b = j["body"]
        if body_jntadr[b] < 0:
            body_jntadr[b], body_dofadr[b] = i, jnt_dofadr[i]
        body_jntnum[b] += 1
        body_dofnum[b] += 6 if j["type"] == JNT_FREE else 1

    dof_parentid = np.full(nv, -1, dtype=np.int32)
    last_dof_of_body = np.full(nbody, -1, dtype=np.int32)  # last dof on the path incl. body itself
    for b in range(1, nbody):
        last = last_dof_of_body[body_parent[b]]
        for d in range(body_dofnum[b]):
            dd = body_dofadr[b] + d
            dof_parentid[dd] = last
            last = dd
        last_dof_of_body[b] = last

    body_rootid = np.zeros(nbody, dtype=np.int32)
    for b in range(1, nbody):
        body_rootid[b] = b if body_parent[b] == 0 else body_rootid[body_parent[b]]

    # --- body inertial frames from geoms --------------------------------------
    body_mass = np.zeros(nbody)
    body_ipos = np.zeros((nbody, 3))
    body_inertia_full = np.zeros((nbody, 3, 3))  # about ipos, body axes
    for b in range(nbody):
        gs = [g for g in geoms if g.body == b and g.gtype != GEOM_PLANE]
        m = sum(g.mass for g in gs)
        if m <= 0:
            continue
        body_mass[b] = m
        body_ipos[b] = sum(g.mass * g.pos for g in gs) / m
        inert = np.zeros((3, 3))
        for g in gs:
            r = quat_to_mat(g.quat)
            ig = r @ np.diag(_geom_inertia(g.gtype, g.size, g.mass)) @ r.T
            d = g.pos - body_ipos[b]
            inert += ig + g.mass * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        body_inertia_full[b] = inert
    body_inertia = np.zeros((nbody, 3))
    body_iquat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
    for b in range(nbody):
        if body_mass[b] <= 0:
            continue
        w, v = np.linalg.eigh(body_inertia_full[b])
        order = np.argsort(-w)  # decreasing, as mju_eig3
        w, v = w[order], v[:, order]
        if np.linalg.det(v) < 0:
            v[:, 2] = -v[:, 2]
        body_inertia[b] = w
        body_iquat[b] = mat_to_quat(v)

    # --- actuators --------------------------------------------------------------
    act_sec = root.find("actuator")
    act_names, act_dof, act_gain, act_tau, act_ctrlrange, act_ctrllimited, act_gear = [], [], [], [], [], [], []
    jnames = [j["name"] for j in joints]
    if act_sec is not None:
        for e in act_sec:
            a = defaults.resolve(e, "general") if e.tag == "general" else defaults.resolve(e)
            if e.tag == "motor":
                gain = 1.0
            else:
                gp = _floats(a.get("gainprm", "1"))
                gain = gp[0]
            if torque_actuators and "forcerange" in a:
                # reference envs/rodent.py:41-45: gainprm=[forcerange[1]], bias removed
                gain = _floats(a["forcerange"])[1]
            jid = jnames.index(a["joint"])
            act_names.append(a.get("name", a["joint"]))
            act_dof.append(jnt_dofadr[jid])
            act_gain.append(gain)
            gear = _floats(a.get("gear", "1"))
            act_gear.append(gear[0])
            dyn = a.get("dyntype", "none")
            act_tau.append(_floats(a.get("dynprm", "1"))[0] if dyn == "filter" else -1.0)
            cr = _floats(a.get("ctrlrange", "0 0"))
            act_ctrlrange.append(cr)
            cl = a.get("ctrllimited", "auto")
            act_ctrllimited.append(1 if (cl == "true" or (cl == "auto" and "ctrlrange" in a)) else 0)
    nu = len(act_names)

    # --- options ------------------------------------------------------------------
    opt = root.find("option")
    timestep = float(opt.attrib.get("timestep", 0.002)) if opt is not None else 0.002
    gravity = _floats(opt.attrib.get("gravity", "0 0 -9.81")) if opt is not None else np.array([0, 0, -9.81])
    eulerdamp = 1
    if opt is not None:
        flag = opt.find("flag")
        if flag is not None and flag.attrib.get("eulerdamp", "enable") == "disable":
            eulerdamp = 0

    arrays: Dict[str, np.ndarray] = dict(
        body_parentid=body_parent,
        body_rootid=body_rootid,
        body_pos=body_pos,
        body_quat=body_quat,
        body_ipos=body_ipos,
        body_iquat=body_iquat,
        body_inertia=body_inertia,
        body_inertia_full=body_inertia_full.reshape(nbody, 9),
        body_mass=body_mass,
        body_jntadr=body_jntadr,
        body_jntnum=body_jntnum,
        body_dofadr=body_dofadr,
        body_dofnum=body_dofnum,
        jnt_type=jnt_type,
        jnt_bodyid=jnt_bodyid,
        jnt_qposadr=jnt_qposadr,
        jnt_dofadr=jnt_dofadr,
        jnt_pos=jnt_pos,
        jnt_axis=jnt_axis,
        jnt_range=jnt_range,
        jnt_limited=jnt_limited,
        jnt_stiffness=jnt_stiffness,
        jnt_margin=jnt_margin,
        jnt_solref=jnt_solref,
        jnt_solimp=jnt_solimp,
        qpos0=qpos0,
        qpos_spring=qpos_spring,
        dof_bodyid=dof_bodyid,
        dof_jntid=dof_jntid,
        dof_parentid=dof_parentid,
        dof_armature=dof_armature,
        dof_damping=dof_damping,
        act_dof=np.array(act_dof, dtype=np.int32),
        act_gain=np.array(act_gain, dtype=np.float64),
        act_gear=np.array(act_gear, dtype=np.float64),
        act_tau=np.array(act_tau, dtype=np.float64),
        act_ctrlrange=np.array(act_ctrlrange, dtype=np.float64).reshape(nu, 2),
        act_ctrllimited=np.array(act_ctrllimited, dtype=np.int32),
        gravity=gravity,
    )
    # all geoms (for inspection / tests)
    arrays.update(
        geom_type=np.array([g.gtype for g in geoms], dtype=np.int32),
        geom_bodyid=np.array([g.body for g in geoms], dtype=np.int32),
        geom_size=np.array([g.size for g in geoms]),
        geom_pos=np.array([g.pos for g in geoms]),
        geom_quat=np.array([g.quat for g in geoms]),
        geom_mass=np.array([g.mass for g in geoms]),
        geom_contype=np.array([g.contype for g in geoms], dtype=np.int32),
        geom_conaffinity=np.array([g.conaffinity for g in geoms], dtype=np.int32),
    )
    scalars = dict(
        nq=nq,
        nv=nv,
        nu=nu,
        na=int(sum(1 for t in act_tau if t >= 0)),
        nbody=nbody,
        njnt=njnt,
        ngeom=len(geoms),
        timestep=timestep,
        tolerance=1e-8,
        ls_tolerance=0.01,
        impratio=float(opt.attrib.get("impratio", 1.0)) if opt is not None else 1.0,
        iterations=iterations,
        ls_iterations=ls_iterations,
        solver_newton=1 if solver.lower() == "newton" else 0,
        eulerdamp=eulerdamp,
    )
    names = dict(
        body=body_names,
        joint=jnames,
        geom=[g.name for g in geoms],
        actuator=act_names,
    )
    model = CompiledModel(names=names, arrays=arrays, scalars=scalars)

    # explicit <contact><pair>: collide whatever contype / conaffinity say (e.g. assets/humanoid.xml:189-197)
    pairs = []
    contact = root.find("contact")
    if contact is not None:
        gname = {g.name: i for i, g in enumerate(geoms)}
        for pe in contact.findall("pair"):
            pa = defaults.resolve(pe, "pair")
            unsupported = [k for k in ("condim", "friction", "solref", "solimp", "margin", "gap") if k in pa]
            if unsupported:
                raise NotImplementedError(f"<pair> attributes {unsupported}: only geom-derived pair parameters are implemented")
            pairs.append((gname[pa["geom1"]], gname[pa["geom2"]]))
    _contact_tables(model, geoms, pairs)
    _set_const(model)
    return model


def _contact_tables(model: CompiledModel, geoms: Sequence[_Geom], explicit_pairs: Sequence[tuple] = ()) -> None:
    """Static geom-vs-plane contact list.

    MJX [UPSTREAM collision_driver] emits a fixed set of contacts per candidate
    geom pair (contype/conaffinity filter): plane-sphere 1, plane-capsule 2,
    plane-ellipsoid 1.  Only plane pairs are supported here; any other candidate
    pair raises, so an unsupported model fails loudly at compile time.
    Contact parameters are mixed as MuJoCo does [engine_collision_driver
    mj_contactParam]: higher priority wins; equal priority -> max friction,
    solmix-weighted solref/solimp, max condim.
    """
    planes = [i for i, g in enumerate(geoms) if g.gtype == GEOM_PLANE]
    rows = []
    explicit = {(min(a, b), max(a, b)) for a, b in explicit_pairs}
    for i, g1 in enumerate(geoms):
        for j in range(i + 1, len(geoms)):
            g2 = geoms[j]
            is_explicit = (i, j) in explicit
            # explicit <pair>s bypass the filters; their unspecified parameters come from the two geoms by the
            # equal-priority mixing rules below [MuJoCo user_objects mjCPair::Compile]
            if not is_explicit:
                if not ((g1.contype & g2.conaffinity) or (g2.contype & g1.conaffinity)):
                    continue
                if g1.body == g2.body:
                    continue
                # parent-child filter (MuJoCo filterparent; world-body parents are exempt)
                bp = model.arrays["body_parentid"]
                if g1.body != 0 and g2.body != 0 and (bp[g1.body] == g2.body or bp[g2.body] == g1.body):
                    continue
            if i in planes:
                p, o = g1, g2
                pi_, oi = i, j
            elif j in planes:
                p, o = g2, g1
                pi_, oi = j, i
            else:
                raise NotImplementedError(f"non-plane collision pair {g1.name} / {g2.name}")
            if o.gtype not in (GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID):
                raise NotImplementedError(f"plane vs geom type {o.gtype}")
            if p.priority > o.priority and not is_explicit:
                fr, sr, si, cd = p.friction, p.solref, p.solimp, p.condim
            elif o.priority > p.priority and not is_explicit:
                fr, sr, si, cd = o.friction, o.solref, o.solimp, o.condim
            else:
                fr = np.maximum(p.friction, o.friction)
                mix = p.solmix / (p.solmix + o.solmix)
                if p.solref[0] > 0 and o.solref[0] > 0:
                    sr = mix * p.solref + (1 - mix) * o.solref
                else:
                    sr = np.minimum(p.solref, o.solref)
                si = mix * p.solimp + (1 - mix) * o.solimp
                cd = max(p.condim, o.condim)
            if cd != 3:
                raise NotImplementedError("only condim 3 contacts")
            rows.append(
                dict(
                    plane=pi_,
                    geom=oi,
                    friction=fr,
                    solref=sr,
                    solimp=si,
                    margin=max(p.margin, o.margin),
                    gap=max(p.gap, o.gap),
                )
            )
    # MJX groups contacts by collision function in geom-type order
    # (plane-sphere, plane-capsule, plane-ellipsoid); within a group by geom id.
    order = {GEOM_SPHERE: 0, GEOM_CAPSULE: 1, GEOM_ELLIPSOID: 2}
    rows.sort(key=lambda r: (order[geoms[r["geom"]].gtype], r["geom"]))
    ng = len(rows)
    a = model.arrays
    a["cg_type"] = np.array([geoms[r["geom"]].gtype for r in rows], dtype=np.int32)
    a["cg_geomid"] = np.array([r["geom"] for r in rows], dtype=np.int32)
    a["cg_bodyid"] = np.array([geoms[r["geom"]].body for r in rows], dtype=np.int32)
    a["cg_pos"] = np.array([geoms[r["geom"]].pos for r in rows]).reshape(ng, 3)
    a["cg_quat"] = np.array([geoms[r["geom"]].quat for r in rows]).reshape(ng, 4)
    a["cg_size"] = np.array([geoms[r["geom"]].size for r in rows]).reshape(ng, 3)
    a["cg_friction"] = np.array([r["friction"] for r in rows]).reshape(ng, 3)
    a["cg_solref"] = np.array([r["solref"] for r in rows]).reshape(ng, 2)
    a["cg_solimp"] = np.array([r["solimp"] for r in rows]).reshape(ng, 5)
    a["cg_margin"] = np.array([r["margin"] - r["gap"] for r in rows]).reshape(ng)
    ncon_per = np.array([2 if geoms[r["geom"]].gtype == GEOM_CAPSULE else 1 for r in rows], dtype=np.int32)
    a["cg_ncon"] = ncon_per
    a["cg_conadr"] = np.concatenate([[0], np.cumsum(ncon_per)[:-1]]).astype(np.int32) if ng else np.zeros(0, np.int32)
    if planes:
        if len(set(r["plane"] for r in rows)) > 1:
            raise NotImplementedError("more than one plane")
        pg = geoms[rows[0]["plane"]] if rows else geoms[planes[0]]
        if pg.body != 0:
            raise NotImplementedError("plane must be on the world body")
        a["plane_pos"] = pg.pos.copy()
        a["plane_normal"] = quat_to_mat(pg.quat)[:, 2].copy()
    else:
        a["plane_pos"] = np.zeros(3)
        a["plane_normal"] = np.array([0.0, 0, 1.0])
    model.scalars["ncg"] = ng
    model.scalars["ncon"] = int(ncon_per.sum())
    nlimit = int(model.arrays["jnt_limited"].sum())
    model.scalars["nlimit"] = nlimit
    model.scalars["nefc"] = nlimit + 4 * int(ncon_per.sum())


# ----------------------------------------------------------------------------
# float64 reference kinematics / mass matrix at a given qpos (host-side, used for
# the qpos0 constants and by clip preprocessing of small clips)
# ----------------------------------------------------------------------------
def forward_kinematics(model: CompiledModel, qpos: np.ndarray):
    """MJX smooth.kinematics restated [UPSTREAM]; returns dict of world-frame arrays."""
    a = model.arrays
    nbody = int(model.scalars["nbody"])
    njnt = int(model.scalars["njnt"])
    xpos = np.zeros((nbody, 3))
    xquat = np.zeros((nbody, 4))
    xquat[0, 0] = 1.0
    xanchor = np.zeros((njnt, 3))
    xaxis = np.zeros((njnt, 3))
    qpos = np.array(qpos, dtype=np.float64)
    for b in range(1, nbody):
        p = a["body_parentid"][b]
        pos = xpos[p] + rotate(a["body_pos"][b], xquat[p])
        quat = quat_mul(xquat[p], a["body_quat"][b])
        for k in range(a["body_jntnum"][b]):
            j = a["body_jntadr"][b] + k
            qa = a["jnt_qposadr"][j]
            if a["jnt_type"][j] == JNT_FREE:
                xanchor[j] = qpos[qa : qa + 3]
                xaxis[j] = [0, 0, 1.0]
                pos = qpos[qa : qa + 3].copy()
                quat = qpos[qa + 3 : qa + 7] / np.linalg.norm(qpos[qa + 3 : qa + 7])
                qpos[qa + 3 : qa + 7] = quat
            else:
                anchor = rotate(a["jnt_pos"][j], quat) + pos
                axis = rotate(a["jnt_axis"][j], quat)
                xanchor[j], xaxis[j] = anchor, axis
                quat = quat_mul(quat, axis_angle_quat(a["jnt_axis"][j], qpos[qa] - a["qpos0"][qa]))
                pos = anchor - rotate(a["jnt_pos"][j], quat)
        xpos[b], xquat[b] = pos, quat / np.linalg.norm(quat)
    xmat = np.array([quat_to_mat(q) for q in xquat])
    xipos = xpos + np.einsum("bij,bj->bi", xmat, a["body_ipos"])
    return dict(qpos=qpos, xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, xanchor=xanchor, xaxis=xaxis)


def subtree_com(model: CompiledModel, xipos: np.ndarray) -> np.ndarray:
    a = model.arrays
    nbody = int(model.scalars["nbody"])
    mp = a["body_mass"][:, None] * xipos
    m = a["body_mass"].copy()
    for b in range(nbody - 1, 0, -1):
        p = a["body_parentid"][b]
        mp[p] += mp[b]
        m[p] += m[b]
    out = xipos.copy()
    nz = m > MJ_MINVAL
    out[nz] = mp[nz] / m[nz, None]
    return out


def mass_matrix(model: CompiledModel, qpos: np.ndarray):
    """Dense joint-space inertia M(qpos) incl. armature, and body-COM Jacobians."""
    a = model.arrays
    nbody, nv = int(model.scalars["nbody"]), int(model.scalars["nv"])
    fk = forward_kinematics(model, qpos)
    xmat, xipos = fk["xmat"], fk["xipos"]
    # world-frame dof axes: columns of the 6D (ang, lin-at-origin) motion subspace
    ang = np.zeros((nv, 3))
    pt = np.zeros((nv, 3))  # a point on the axis (for hinges), unused for translations
    is_trans = np.zeros(nv, dtype=bool)
    for j in range(int(model.scalars["njnt"])):
        da = a["jnt_dofadr"][j]
        b = a["jnt_bodyid"][j]
        if a["jnt_type"][j] == JNT_FREE:
            for k in range(3):
                is_trans[da + k] = True
                ang[da + k, k] = 1.0  # direction stored in ang for translations
            for k in range(3):
                ang[da + 3 + k] = xmat[b][:, k]
                pt[da + 3 + k] = fk["xpos"][b]
        else:
            ang[da] = fk["xaxis"][j]
            pt[da] = fk["xanchor"][j]

    def jac_point(body: int, point: np.ndarray):
        jp_, jr = np.zeros((3, nv)), np.zeros((3, nv))
        d = a["body_dofadr"][body] + a["body_dofnum"][body] - 1 if a["body_dofnum"][body] > 0 else -1
        if d < 0:
            # climb to nearest ancestor with dofs
            bb = body
            while bb > 0 and a["body_dofnum"][bb] == 0:
                bb = a["body_parentid"][bb]
            d = a["body_dofadr"][bb] + a["body_dofnum"][bb] - 1 if bb > 0 else -1
        while d >= 0:
            if is_trans[d]:
                jp_[:, d] = ang[d]
            else:
                jr[:, d] = ang[d]
                jp_[:, d] = np.cross(ang[d], point - pt[d])
            d = a["dof_parentid"][d]
        return jp_, jr

    M = np.zeros((nv, nv))
    jacs = []
    for b in range(nbody):
        if b == 0:
            jacs.append((np.zeros((3, nv)), np.zeros((3, nv))))
            continue
        jp_, jr = jac_point(b, xipos[b])
        jacs.append((jp_, jr))
        if a["body_mass"][b] > 0:
            iw = xmat[b] @ a["body_inertia_full"][b].reshape(3, 3) @ xmat[b].T
            M += a["body_mass"][b] * jp_.T @ jp_ + jr.T @ iw @ jr
    M += np.diag(a["dof_armature"])
    return M, jacs, fk


def _set_const(model: CompiledModel) -> None:
    """dof_invweight0 / body_invweight0 / meaninertia at qpos0 [MuJoCo engine_setconst set0]."""
    a = model.arrays
    nbody, nv = int(model.scalars["nbody"]), int(model.scalars["nv"])
    M, jacs, fk = mass_matrix(model, a["qpos0"])
    Minv = np.linalg.inv(M)
    model.scalars["meaninertia"] = float(np.trace(M) / max(nv, 1))
    biw = np.zeros((nbody, 2))
    for b in range(1, nbody):
        jp_, jr = jacs[b]
        biw[b, 0] = max(MJ_MINVAL, np.trace(jp_ @ Minv @ jp_.T) / 3.0)
        biw[b, 1] = max(MJ_MINVAL, np.trace(jr @ Minv @ jr.T) / 3.0)
    diw = np.zeros(nv)
    for j in range(int(model.scalars["njnt"])):
        da = a["jnt_dofadr"][j]
        if a["jnt_type"][j] == JNT_FREE:
            diw[da : da + 3] = np.mean(np.diag(Minv)[da : da + 3])
            diw[da + 3 : da + 6] = np.mean(np.diag(Minv)[da + 3 : da + 6])
        else:
            diw[da] = Minv[da, da]
    a["body_invweight0"] = biw
    a["dof_invweight0"] = diw
    a["body_subtreemass"] = _subtree_mass(model)
    a["xpos0"] = fk["xpos"]


def _subtree_mass(model: CompiledModel) -> np.ndarray:
    a = model.arrays
    m = a["body_mass"].copy()
    for b in range(int(model.scalars["nbody"]) - 1, 0, -1):
        m[a["body_parentid"][b]] += m[b]
    return m

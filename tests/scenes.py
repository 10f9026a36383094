"""Seeded test scenes shared by the oracle and the HIP path (inputs only -- no rendering here)."""
import math
import os

import numpy as np


class SceneData:
    def __init__(self):
        self.meshes = []      # dict(verts (n,3) f32, uvs (n,2) f32, idx (m,3) i32, material int)
        self.materials = []   # dict(kd, ks, ns, tex_kd, tex_ks, tex_ns)
        self.textures = []    # (h, w, 4) float32
        self.light_mesh = -1
        self.light_intensity = [17.0, 12.0, 4.0, 0.0]
        self.cam_origin = [0.0, 0.0, 0.0]
        self.cam_lookat = [0.0, 1.0, 0.0]
        self.cam_up = [0.0, 0.0, 1.0]
        self.fovy = math.radians(60.0)
        self.aspect = 1.0

    # ---- construction helpers
    def add_material(self, kd, ks=(0, 0, 0), ns=0.0, tex_kd=-1, tex_ks=-1, tex_ns=-1):
        self.materials.append(dict(kd=[float(x) for x in kd], ks=[float(x) for x in ks], ns=float(ns),
                                   tex_kd=tex_kd, tex_ks=tex_ks, tex_ns=tex_ns))
        return len(self.materials) - 1

    def add_mesh(self, verts, idx, material, uvs=None):
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx, dtype=np.int32).reshape(-1, 3)
        uvs = np.zeros((verts.shape[0], 2), np.float32) if uvs is None else np.ascontiguousarray(uvs, dtype=np.float32).reshape(-1, 2)
        self.meshes.append(dict(verts=verts, uvs=uvs, idx=idx, material=material))
        return len(self.meshes) - 1

    def add_quad(self, o, u, v, material, nu=1, nv=1):
        """Tessellated parallelogram with normal along u x v."""
        o, u, v = (np.asarray(a, dtype=np.float32) for a in (o, u, v))
        s, t = np.meshgrid(np.arange(nu + 1, dtype=np.float32) / nu, np.arange(nv + 1, dtype=np.float32) / nv)
        verts = o[None, None, :] + s[..., None] * u[None, None, :] + t[..., None] * v[None, None, :]
        uvs = np.stack([s, t], axis=-1)
        idx = []
        for j in range(nv):
            for i in range(nu):
                a = j * (nu + 1) + i; b = a + 1; c = b + nu + 1; d = a + nu + 1
                idx += [(a, b, c), (a, c, d)]
        return self.add_mesh(verts.reshape(-1, 3), idx, material, uvs.reshape(-1, 2))

    def add_box(self, lo, hi, material, outward=True, n=1):
        lo = np.asarray(lo, np.float32); hi = np.asarray(hi, np.float32); d = hi - lo
        ids = []
        for axis in range(3):
            for side in range(2):
                a1, a2 = (axis + 1) % 3, (axis + 2) % 3
                o = lo.copy(); u = np.zeros(3, np.float32); v = np.zeros(3, np.float32)
                if side:
                    o[axis] = hi[axis]
                if (side == 1) == outward:
                    u[a1] = d[a1]; v[a2] = d[a2]
                else:
                    u[a2] = d[a2]; v[a1] = d[a1]
                ids.append(self.add_quad(o, u, v, material, n, n))
        return ids

    def merge_meshes(self, ids, material):
        """Concatenate several meshes into one (e.g. the single light mesh)."""
        verts, uvs, idx, base = [], [], [], 0
        for i in ids:
            m = self.meshes[i]
            verts.append(m["verts"]); uvs.append(m["uvs"]); idx.append(m["idx"] + base); base += m["verts"].shape[0]
        keep = [m for k, m in enumerate(self.meshes) if k not in ids]
        self.meshes = keep
        return self.add_mesh(np.concatenate(verts), np.concatenate(idx), material, np.concatenate(uvs))

    # ---- views
    def triangle_soup(self):
        """Flattened (verts9, uv6, material) in mesh order, the light mesh keeps its place."""
        V, U, M = [], [], []
        first = 0
        for k, m in enumerate(self.meshes):
            v = m["verts"][m["idx"]].reshape(-1, 9); u = m["uvs"][m["idx"]].reshape(-1, 6)
            if k == self.light_mesh:
                self.light_first, self.light_count = first, v.shape[0]
            V.append(v); U.append(u); M.append(np.full(v.shape[0], m["material"], np.int32)); first += v.shape[0]
        return (np.ascontiguousarray(np.concatenate(V), np.float32), np.ascontiguousarray(np.concatenate(U), np.float32),
                np.ascontiguousarray(np.concatenate(M), np.int32))

    def upload(self, ctx):
        """Feed an evplp_amd.Context through the C ABI (textures, materials, meshes, light, camera, accel)."""
        tex_ids = [ctx.add_texture(t) for t in self.textures]
        mat_ids = []
        for m in self.materials:
            mat_ids.append(ctx.add_material(m["kd"], m["ks"], m["ns"],
                                            tex_ids[m["tex_kd"]] if m["tex_kd"] >= 0 else -1,
                                            tex_ids[m["tex_ks"]] if m["tex_ks"] >= 0 else -1,
                                            tex_ids[m["tex_ns"]] if m["tex_ns"] >= 0 else -1))
        light = -1
        for k, m in enumerate(self.meshes):
            mid = ctx.add_mesh(m["verts"], m["idx"], mat_ids[m["material"]], m["uvs"])
            if k == self.light_mesh:
                light = mid
        ctx.set_arealight(light, self.light_intensity)
        ctx.set_camera(self.cam_origin, self.cam_lookat, self.cam_up, self.fovy, self.aspect)
        ctx.build_accel()


def box_room(seed=1, n_boxes=6, tess=2, glossy=True, textured=False, aspect=1.0):
    """Closed room [0,10]x[0,8]x[0,5] with random boxes, one ceiling light (2 quads merged), a camera inside."""
    rng = np.random.RandomState(seed)
    s = SceneData()
    s.aspect = aspect
    tex = -1
    if textured:
        t = np.zeros((8, 8, 4), np.float32)
        t[..., :3] = 0.25 + 0.5 * rng.rand(8, 8, 3).astype(np.float32)
        s.textures.append(t)
        tex = 0
    wall = [s.add_material(0.2 + 0.6 * rng.rand(3)) for _ in range(5)]
    floor = s.add_material(0.2 + 0.6 * rng.rand(3), tex_kd=tex)
    mats = wall + [floor]
    # room shell, normals inward: one material per face
    lo, hi = np.array([0, 0, 0], np.float32), np.array([10, 8, 5], np.float32)
    d = hi - lo
    k = 0
    for axis in range(3):
        for side in range(2):
            a1, a2 = (axis + 1) % 3, (axis + 2) % 3
            o = lo.copy(); u = np.zeros(3, np.float32); v = np.zeros(3, np.float32)
            if side:
                o[axis] = hi[axis]
            if side == 0:
                u[a1] = d[a1]; v[a2] = d[a2]
            else:
                u[a2] = d[a2]; v[a1] = d[a1]
            m = mats[5] if (axis == 2 and side == 0) else mats[k % 5]
            s.add_quad(o, u, v, m, tess * 2, tess * 2)
            k += 1
    for b in range(n_boxes):
        c = np.array([1.5 + 7.0 * rng.rand(), 1.5 + 5.0 * rng.rand(), 0.0])
        sz = np.array([0.4 + 1.2 * rng.rand(), 0.4 + 1.2 * rng.rand(), 0.5 + 2.5 * rng.rand()])
        ks = (0.2, 0.2, 0.2) if (glossy and b % 3 == 2) else (0, 0, 0)
        ns = 20.0 if (glossy and b % 3 == 2) else 0.0
        m = s.add_material(0.2 + 0.6 * rng.rand(3), ks, ns)
        s.add_box(c - [sz[0] / 2, sz[1] / 2, 0], c + [sz[0] / 2, sz[1] / 2, sz[2]], m, True, tess)
    # light: two quads facing down, merged into one mesh
    lm = s.add_material((0, 0, 0))
    a = s.add_quad([3.0, 3.0, 4.9], [0, 1.5, 0], [1.5, 0, 0], lm, 2, 2)
    b = s.add_quad([6.0, 3.5, 4.9], [0, 1.0, 0], [2.0, 0, 0], lm, 1, 1)
    s.light_mesh = s.merge_meshes([a, b], lm)
    s.light_intensity = [17.0, 12.0, 4.0, 0.0]
    s.cam_origin = [9.2, 0.9, 3.2]; s.cam_lookat = [3.0, 5.0, 2.9]; s.cam_up = [0.0, 0.0, 1.0]
    s.fovy = math.radians(55.0)
    s.triangle_soup()
    return s


def load_obj_scene(json_path, decode=None):
    """Independent Python reading of a scene JSON + OBJ/MTL (same semantics as csrc/host/scene_io.cpp).
    decode(path) -> uint8 (h, w, 3) top-down pixels for map_Kd / map_Ks / map_Ns (RtTexture(filepath, 1.0),
    rtcommon.h:139-194: byte / 255, flipped vertically)."""
    import json
    root = json.load(open(json_path))
    d = os.path.dirname(json_path)
    s = SceneData()

    def read_mtl(path):
        out, cur = {}, None
        if not os.path.exists(path):
            return out
        for line in open(path):
            t = line.split()
            if not t:
                continue
            if t[0] == "newmtl":
                cur = t[1]; out[cur] = dict(kd=[0.6, 0.6, 0.6], ks=[0, 0, 0], ns=0.0)
            elif cur and t[0] == "Kd":
                out[cur]["kd"] = [float(x) for x in t[1:4]]
            elif cur and t[0] == "Ks":
                out[cur]["ks"] = [float(x) for x in t[1:4]]
            elif cur and t[0] == "Ns":
                out[cur]["ns"] = float(t[1])
            elif cur and t[0] in ("map_Kd", "map_Ks", "map_Ns"):
                out[cur][t[0]] = os.path.join(os.path.dirname(path), t[1])
        return out

    tex_ids = {}

    def tex_of(m, key):
        if key not in m:
            return -1
        if m[key] not in tex_ids:
            px = decode(m[key])
            t = np.zeros(px.shape[:2] + (4,), np.float32)
            t[..., :3] = px[::-1].astype(np.float32) / np.float32(255.0)
            tex_ids[m[key]] = len(s.textures); s.textures.append(t)
        return tex_ids[m[key]]

    def read_obj(path, want_materials):
        pos, tex, mtl = [], [], {}
        names, groups = [""], [dict(verts=[], uvs=[], idx=[], map={})]
        cur = 0
        for line in open(path):
            if line.startswith("v "):
                pos.append([np.float32(x) for x in line.split()[1:4]])
            elif line.startswith("vt"):
                tex.append([np.float32(x) for x in line.split()[1:3]])
            elif line.startswith("f "):
                face = []
                for tok in line.split()[1:]:
                    p = tok.split("/")
                    vi = int(p[0]); ti = int(p[1]) if len(p) > 1 and p[1] else 0
                    vi = vi - 1 if vi > 0 else len(pos) + vi
                    ti = (ti - 1 if ti > 0 else len(tex) + ti) if ti != 0 else -1
                    face.append((vi, ti))
                g = groups[cur]

                def idx_of(k):
                    if k not in g["map"]:
                        g["map"][k] = len(g["verts"])
                        g["verts"].append(pos[k[0]]); g["uvs"].append(tex[k[1]] if k[1] >= 0 else [0.0, 0.0])
                    return g["map"][k]
                for k in range(1, len(face) - 1):
                    g["idx"].append([idx_of(face[0]), idx_of(face[k]), idx_of(face[k + 1])])
            elif line.startswith("usemtl"):
                name = line.split()[1]
                if name not in names:
                    names.append(name); groups.append(dict(verts=[], uvs=[], idx=[], map={}))
                cur = names.index(name)
            elif line.startswith("mtllib"):
                mtl.update(read_mtl(os.path.join(os.path.dirname(path), line.split()[1])))
        res = []
        for name, g in zip(names, groups):
            if not g["idx"]:
                continue
            m = mtl.get(name, dict(kd=[0.6, 0.6, 0.6], ks=[0, 0, 0], ns=0.0))
            res.append((g, m))
        return res

    for rel in root["scene"]:
        for g, m in read_obj(os.path.join(d, rel), True):
            mid = s.add_material(m["kd"], m["ks"], m["ns"], tex_of(m, "map_Kd"), tex_of(m, "map_Ks"), tex_of(m, "map_Ns"))
            s.add_mesh(np.array(g["verts"], np.float32), np.array(g["idx"], np.int32), mid, np.array(g["uvs"], np.float32))
    lg = read_obj(os.path.join(d, root["arealight"]["obj"]), False)
    assert len(lg) == 1
    lm = s.add_material((0, 0, 0))
    s.light_mesh = s.add_mesh(np.array(lg[0][0]["verts"], np.float32), np.array(lg[0][0]["idx"], np.int32), lm, np.array(lg[0][0]["uvs"], np.float32))
    s.light_intensity = [float(x) for x in root["arealight"]["intensity"]]
    cam = root.get("camera", root.get("stablecamera"))
    s.aspect = float(np.float32(root["resX"]) / np.float32(root["resY"]))
    s.cam_origin, s.cam_lookat, s.cam_up = cam["origin"], cam["direction"], cam["up"]
    deg = np.float32(0.01745329251994329576923690768489)
    if "fovy" in cam:
        s.fovy = float(np.float32(cam["fovy"]) * deg)
    else:
        s.fovy = float(np.float32(2.0) * np.arctan2(np.tan(np.float32(cam["fovx"]) * deg * np.float32(0.5)), np.float32(s.aspect)))
    s.triangle_soup()
    return s, root

"""Loader for the golden fixtures (tests/golden/*.npz, written by tests/golden/generate_golden.py)."""
import glob
import hashlib
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NAN_CODE, INF_CODE = 65535, 65534


def fixture_paths():
    return sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def fixture_names():
    return [os.path.splitext(os.path.basename(p))[0] for p in fixture_paths()]


class SeededInputs:
    """The inputs of one stream of a bench workload, rebuilt from a recipe (tests/golden/generate_seeded.py): the arguments of
    bench_support.configs.build, the rank-local stream and the step.  Deterministic: seeds derive from global stream / URDF numbers."""

    def __init__(self, recipe):
        from bench_support import configs as CF, synthetic
        share = CF.build(**recipe["build"])
        s, k = int(recipe["stream"]), int(recipe["step"])
        # model ids are only known once a context has loaded the share; the oracle's view needs none of them
        for g in share.groups:
            g.model_ids = list(range(len(g.variants[0].models)))
        base = 0
        for g in share.groups:
            for m, links in zip(g.model_ids, g.variants[0].models):
                share.link_base[m] = base
                base += len(links)
        self.projection, self.draws, self.offset_inv, self.cam_tf = share.oracle_frame(k, s)
        wl = share.wl0
        self.width, self.height = share.width, share.height
        self.near, self.far = wl.near, wl.far
        self.max_diff, self.replace_value = float(wl.max_diff), float(wl.replace_value)
        self.depth = synthetic.sensor_depth(self.width, self.height, int(recipe["depth_seed"]))
        self.triangles = int(sum(len(d[4]) for d in self.draws))
        h = hashlib.sha256()
        for a in (self.depth, self.projection, self.offset_inv, self.cam_tf):
            h.update(np.ascontiguousarray(a).tobytes())
        for tf, pre, op, v, t in self.draws:
            h.update(np.ascontiguousarray(tf, np.float64).tobytes())
            h.update(np.asarray([pre], np.int32).tobytes() + np.asarray(op, np.float32).tobytes())
            h.update(np.ascontiguousarray(v, np.float32).tobytes() + np.ascontiguousarray(t, np.uint32).tobytes())
        self.inputs_sha256 = h.digest()


def seeded_inputs(recipe):
    return SeededInputs(recipe)


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(HERE, "golden", name + ".npz"))
        self.name = name
        if "recipe" in z.files:          # inputs addressed by seed (BASELINE-size frames): rebuilt, and checked against the hash taken when the reference rendered them
            import json
            fi = SeededInputs(json.loads(z["recipe"].tobytes().decode()))
            assert fi.inputs_sha256 == bytes(z["inputs_sha256"].tobytes()), \
                "%s: the workload generator no longer produces the inputs this fixture was rendered from (regenerate: tests/golden/generate_seeded.py)" % name
            assert (fi.width, fi.height) == (int(z["width"]), int(z["height"]))
            self.width, self.height = fi.width, fi.height
            self.max_diff, self.replace_value = fi.max_diff, fi.replace_value
            self.projection, self.offset_inv, self.cam_tf, self.depth = fi.projection, fi.offset_inv, fi.cam_tf, fi.depth
            self.draws = [(np.asarray(tf, np.float64), int(pre), [float(x) for x in op], v, t) for tf, pre, op, v, t in fi.draws]
            self.mask = (np.unpackbits(z["mask_bits"])[: self.width * self.height].reshape(self.height, self.width) * 255).astype(np.uint8)
            self.masked_sha256 = bytes(z["masked_sha256"].tobytes())
            self.renderer = z["renderer"].tobytes().decode()
            return
        self.width, self.height = int(z["width"]), int(z["height"])
        self.max_diff, self.replace_value = float(z["max_diff"]), float(z["replace_value"])
        self.projection, self.offset_inv, self.cam_tf = z["projection"], z["offset_inv"], z["cam_tf"]
        if "depth_q" in z.files:
            q = z["depth_q"]
            d = (q.astype(np.float32) / np.float32(1024.0)).astype(np.float32)
            d[q == NAN_CODE] = np.nan
            d[q == INF_CODE] = np.inf
            self.depth = d
        else:
            self.depth = z["depth_f32"]
        self.draws = []
        vo = to = 0
        for i in range(len(z["pre_op"])):
            nv, nt = int(z["vert_count"][i]), int(z["tri_count"][i])
            self.draws.append((z["link_tf"][i], int(z["pre_op"][i]), [float(x) for x in z["op"][i]],
                               z["verts"][vo:vo + nv], z["tris"][to:to + nt]))
            vo += nv
            to += nt
        self.mask = (np.unpackbits(z["mask_bits"])[: self.width * self.height].reshape(self.height, self.width) * 255).astype(np.uint8)
        self.masked_sha256 = bytes(z["masked_sha256"].tobytes())
        self.renderer = z["renderer"].tobytes().decode()

    def expected_masked(self):
        """The reference's colour attachment 1: replace value where filtered, the sensor value elsewhere."""
        m = np.where(self.mask > 0, np.float32(self.replace_value), self.depth).astype(np.float32)
        assert hashlib.sha256(m.tobytes()).digest() == self.masked_sha256
        return m

    def check(self, masked, mask):
        assert mask.shape == self.mask.shape
        bad = int((mask != self.mask).sum())
        assert bad == 0, "%s: %d mask pixels differ from the reference" % (self.name, bad)
        assert hashlib.sha256(np.ascontiguousarray(masked, np.float32).tobytes()).digest() == self.masked_sha256, \
            "%s: masked depth differs from the reference" % self.name

    def load_into(self, ctx):
        """One model, one link + one draw per fixture draw; returns (model id, link_tf [L,16])."""
        m = ctx.add_model()
        for tf, pre, op, v, t in self.draws:
            l = ctx.add_link(m)
            ctx.add_draw(m, l, v, t, pre, op)
        ctx.finalize_models()
        tfs = np.stack([d[0] for d in self.draws]) if self.draws else np.zeros((0, 16))
        return m, tfs

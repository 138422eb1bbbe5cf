"""BSDF restatement: energy and pdf checks on the reference's material table (main.cu:443-467)."""
import numpy as np

from conftest import golden_scene


def _wi(theta, phi=0.3):
    # local direction INTO the surface (the convention of sample_f_eval's wi argument)
    return -np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], np.float32)


def test_lambert_white_furnace(oracle):
    sc = oracle.OracleScene(golden_scene("cornell32"))
    for mat, albedo in ((2, (0.9, 0.9, 0.9)), (6, (0.9, 0.1, 0.1)), (23, (0.1, 0.9, 0.1))):
        for k in range(200):
            o = sc.bsdf_sample(mat, _wi(0.7), subseq=k)
            wo, f, pdf, draws = o[0:3], o[3:6], o[6], o[7]
            assert draws == 2 and wo[2] > 0 and abs(np.linalg.norm(wo) - 1) < 1e-5
            # f*cos/pdf == albedo for cosine sampling (unless the pdf floor max(z, eps) bites)
            assert np.allclose(f * wo[2] / pdf, albedo, rtol=2e-5)
            ev = sc.bsdf_eval(mat, _wi(0.7), wo)
            assert np.array_equal(ev[:3], f) and ev[3] == pdf


def test_mirror_and_dielectric(oracle):
    sc = oracle.OracleScene(golden_scene("mixed32"))
    wi = _wi(0.5)
    o = sc.bsdf_sample(19, wi, subseq=1)
    assert o[7] == 0 and np.allclose(o[0:3], [wi[0], wi[1], -wi[2]]) and o[6] == 1.0
    assert np.allclose(o[3:6] * o[2], 1.0)          # f * cos = 1
    n_refl = n_refr = 0
    for k in range(400):
        o = sc.bsdf_sample(5, wi, backface=False, subseq=k)
        assert o[7] == 1
        if o[2] > 0:
            n_refl += 1
            assert np.allclose(o[3:6] * o[2] / o[6], 1.0, rtol=1e-5)
        else:
            n_refr += 1
            eta = 1.0 / 1.5
            assert np.allclose(o[3:6] * abs(o[2]) / o[6], eta * eta, rtol=1e-5)
    assert n_refr > n_refl > 0
    # total internal reflection from inside at a grazing angle: no draw, pdf 1
    o = sc.bsdf_sample(5, _wi(1.3), backface=True, subseq=3)
    assert o[7] == 0 and o[6] == 1.0 and o[2] > 0


def test_ggx_metal(oracle):
    sc = oracle.OracleScene(golden_scene("metal32"))
    wi = _wi(0.6)
    vals = []
    for k in range(300):
        o = sc.bsdf_sample(7, wi, subseq=k)
        assert o[7] == 2 and o[2] > 0
        ev = sc.bsdf_eval(7, wi, o[0:3])
        assert np.array_equal(ev[:3], o[3:6]) and ev[3] == o[6]
        if o[6] > 0:
            vals.append(o[3:6] * o[2] / o[6])
    m = np.mean(vals, axis=0)
    assert np.all(m > 0.2) and np.all(m < 1.3)      # Fresnel-weighted reflectance of a rough conductor


def test_unhandled_material_types_leave_outputs(oracle):
    sc = oracle.OracleScene(golden_scene("cornell32"))
    ev = sc.bsdf_eval(5, _wi(0.4), np.array([0, 0, 1], np.float32))     # dielectric: f_eval writes nothing, pdf 0
    assert np.array_equal(ev, [0, 0, 0, 0])

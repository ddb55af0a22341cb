"""The drop-in `src` package against the module surface the reference's driver and modules rely on.

tests/golden/surface.json (written by tests/golden/make_surface.py from a parse of the reference's files: names,
arities and literal defaults only) lists every function the reference's `src` modules define and every
`module.attribute` use in custom_slam.py and src/*.py.  No GPU is needed: nothing here calls into the library.
"""
import importlib
import inspect
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SURFACE = json.load(open(os.path.join(HERE, "golden", "surface.json")))


def _module(name):
    return importlib.import_module("src." + name)


def _defs():
    return [(m, f) for m, fs in SURFACE["defs"].items() for f in fs]


@pytest.mark.parametrize("module,func", _defs())
def test_every_reference_function_exists_with_its_parameters(module, func):
    ref = SURFACE["defs"][module][func]
    ours = getattr(_module(module), func)
    assert callable(ours)
    sig = inspect.signature(ours)
    params = list(sig.parameters.values())
    if ref["varargs"]:
        assert any(p.kind is p.VAR_POSITIONAL for p in params)
        return
    names = [p.name for p in params if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    # the reference's parameters come first, under the same names (extra trailing parameters must be optional)
    assert names[:len(ref["params"])] == ref["params"], (names, ref["params"])
    for p in params[len(ref["params"]):]:
        assert p.default is not p.empty or p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD), f"extra required parameter {p.name}"
    # the same parameters are optional, with the same literal defaults
    n_required = len(ref["params"]) - ref["n_defaults"]
    for i, p in enumerate(params[:len(ref["params"])]):
        if i < n_required:
            assert p.default is p.empty, f"{p.name} must be required"
        else:
            d = ref["defaults"][i - n_required]
            assert p.default is not p.empty, f"{p.name} must have a default"
            if d["literal"]:
                ours_default = list(p.default) if isinstance(p.default, tuple) else p.default
                assert ours_default == d["value"], (p.name, p.default, d["value"])


def _uses():
    out = []
    for f, entries in SURFACE["uses"].items():
        for key, e in entries.items():
            out.append((f, key))
    return out


@pytest.mark.parametrize("where,key", _uses())
def test_every_use_in_the_reference_binds(where, key):
    module, attr = key.split(".")
    target = getattr(_module(module), attr)       # the attribute exists
    for call in SURFACE["uses"][where][key]["calls"]:
        if call["star"]:
            continue
        sig = inspect.signature(target)
        sig.bind(*([None] * call["n_positional"]), **{k: None for k in call["keywords"]})   # raises TypeError if not


def test_plot_and_display_calls_return_without_raising(capsys):
    import numpy as np
    from src import display, logger, plot
    logger.set_log_level("ERROR")
    try:
        img = np.zeros((4, 4), dtype=np.uint8)
        pts = np.zeros((2, 2))
        # the calls custom_slam.py makes (plot: line 77; display: lines 184-198)
        assert plot.plot_image_matches(img, pts, img, pts, subplots=None, show=False) is None
        plot.plot_image_keypoints(img, pts)
        plot.plot_vectors([])
        plot.plot_vectors_with_starting_point({})
        display.setup()
        assert display.should_quit() is True      # the viewer loop of the driver must end at once
        display.init_frame()
        display.draw_camera(np.eye(4), (0.0, 1.0, 0.0))
        display.draw_camera(np.eye(4))
        display.draw_points(np.zeros((3, 3)))
        display.finish_frame()
    finally:
        logger.set_log_level("DEBUG")
    assert capsys.readouterr().out == ""


def test_stand_ins_say_so_once(capsys):
    from src import logger, plot
    plot._told.clear()
    logger.set_log_level("WARN")
    try:
        plot.plot_vectors([])
        plot.plot_vectors([])
    finally:
        logger.set_log_level("DEBUG")
    out = capsys.readouterr().out
    assert out.count("plot_vectors") == 1


def test_util_helpers_match_their_definitions():
    import numpy as np
    from src import util
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    t = rng.normal(size=3)
    pose = util.rotation_translation_to_pose(q, t)
    assert np.allclose(util.pose_to_rotation(pose), q) and np.allclose(util.pose_to_translation(pose), t)
    assert np.allclose(util.vector_to_cross_product_matrix(t) @ t, 0)
    assert util.to_homogenous_coordinates(np.zeros((5, 2))).shape == (5, 3)
    assert util.point_distance(np.array([0.0, 3.0]), np.array([4.0, 0.0])) == 5.0
    sp = pytest.importorskip("sympy")
    x, y = sp.symbols("x y")
    got = util.collect_symbolic_equation_with_respect_to_vars((x + 2 * y) ** 2, [x, y])
    assert got == {x**2: 1, x * y: 4, y**2: 4}

"""driver.LazyArtifacts: the lazily filled dictionaries build_bases / compress_all_parameters hand out must behave as
dictionaries from every side, including the C-level paths that bypass the Python wrappers."""
import pickle

from svdq_amd.driver import LazyArtifacts


def _make(calls):
    def fill():
        calls.append(1)
        return {"a": 1, "b": {"c": 2}}
    return LazyArtifacts(fill)


def test_two_unfilled_instances_compare_equal_and_fill_once():
    c1, c2 = [], []
    x, y = _make(c1), _make(c2)
    assert x == y and not (x != y)                     # neither was filled before the comparison
    assert {"a": 1, "b": {"c": 2}} == y and x == {"a": 1, "b": {"c": 2}}
    assert len(c1) == 1 and len(c2) == 1


def test_unfilled_on_the_right_hand_side_of_dict_operations():
    y = _make([])
    assert {"a": 1, "b": {"c": 2}} == y                # dict.__eq__(plain, lazy): reflected __eq__ fills
    z = _make([])
    merged = {"k": 0} | z                              # dict.__or__ reads z through the C API
    assert merged == {"k": 0, "a": 1, "b": {"c": 2}}
    d = {"k": 0}
    d.update(_make([]))
    assert d == merged
    w = _make([])
    w |= _make([])
    assert dict(w) == {"a": 1, "b": {"c": 2}}


def test_clear_on_unfilled_is_not_undone_and_reversed_works():
    x = _make([])
    x.clear()
    assert len(x) == 0 and list(x) == []
    y = _make([])
    assert list(reversed(y)) == ["b", "a"]
    assert pickle.loads(pickle.dumps(_make([]))) == {"a": 1, "b": {"c": 2}}

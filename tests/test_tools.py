"""tools/source_hash.py: the stamp of the committed counter files (profiles/counters_*.json) follows what the compiler sees."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from source_hash import kernel_source_hash, strip_comments_and_space  # noqa: E402


def test_comments_and_white_space_do_not_count():
    a = 'int f(int x) { return x + 1; }  // adds one\n'
    b = '/* adds\n one */ int f(int x)\n{\n    return x + 1;   /* here */\n}\n'
    assert strip_comments_and_space(a).strip() == strip_comments_and_space(b).strip()


def test_tokens_and_literals_count():
    a = 'const char* s = "a // b"; int k = 1;'
    assert strip_comments_and_space(a) == a
    assert strip_comments_and_space('int k = 1;') != strip_comments_and_space('int k = 2;')
    assert strip_comments_and_space("char c = '\\''; // x") == "char c = '\\''; "
    assert strip_comments_and_space('s = "x\\"//y";') == 's = "x\\"//y";'


def test_committed_counters_carry_a_stamp():
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "profiles", "counters_*.json"))
    assert files
    for f in files:
        d = json.load(open(f))
        assert len(d["kernel_source_hash"]) == 16 and d["workload"] in os.path.basename(f)
    assert len(kernel_source_hash()) == 16

"""Writes tests/golden/reference_tests_cpp.json and survey_probed_cases.json.

These are DATA transcribed from the reference's own test file
(/root/reference/tests/tests.cpp:137-217, known-answer calls `check(text, vocab,
expected)`; unknown id = -1 because none of those vocabularies holds "[UNK]") and
from SURVEY.md §0.2 (outputs the survey recorded from the reference's public
API).  `expected: null` = the reference only asserts Linear == Fast there.
Run:  python tests/golden/make_reference_vectors.py
"""
import json
import os

U = -1
tests_cpp = [
    # testSimple  tests.cpp:137-162
    ("aaaa", ["aaaa", "aaa", "aa", "a"], None),
    ("abcdef", ["bcde", "ac", "def", "bc", "bcdef", "a"], [U]),
    ("abcdef", ["bcde", "ac", "def", "bc", "##bcdef", "a"], [5, 4]),
    ("   aaaa  ", ["aa", "##aa"], [0, 1]),
    ("   aaaa  ", ["aa"], [U]),
    ("aaaa", ["aaaa"], [0]),
    ("aaaa", ["##aaaa"], [U]),
    ("aaaa", ["aaaa", "##aaaa", "##aaa", "##aa", "##a"], [0]),
    ("aaaa", ["##aaa", "aaaa", "##aa", "##a"], [1]),
    ("aaaa", ["aaa", "##aa", "##a", "##aaa"], [0, 2]),
    ("aaaa", ["aa", "a", "##aa"], [0, 2]),
    ("aaaa", ["aa", "a", "##aaa"], [U]),
    ("aaaa", ["aa", "##a"], [0, 1, 1]),
    ("abcdef", ["##def", "abc"], [1, 0]),
    ("abcdef", ["##bcde", "##ac", "##def", "##bc", "##bcdef", "a", "##a"], [5, 4]),
    ("abcdef", ["##bcdd", "##ac", "##def", "##bc", "##bcdff", "a"], [5, 3, 2]),
    ("djzhoyuhmcij", ["d", "##j", "##z", "##h", "##o", "##y", "##u", "##m", "##c", "##i", "##d"],
     [0, 1, 2, 3, 4, 5, 6, 3, 7, 8, 9, 1]),
    # testNonSplitted  tests.cpp:170-176
    ("abc", ["a", "abd"], [U]),
    ("abc a abc abd", ["a", "abd"], [U, 0, U, 1]),
    ("abcdef", ["bcde", "ac", "def", "bc", "bcdef", "##a", "##b", "##c", "##d"], [U]),
    # testPunctuation  tests.cpp:164-168
    ("self-made", ["self", "made", "-", "##-", "##made"], [0, 2, 1]),
    ("self, made", ["self", "made", ",", "##,", "##made"], [0, 2, 1]),
    ("self  , made", ["self", "made", ",", "##,", "##made"], [0, 2, 1]),
    # testMaxMatch  tests.cpp:178-206
    ("abcdef", ["a", "##bcdef", "ab", "##c", "##d", "##e", "##f"], [2, 3, 4, 5, 6]),
    ("abcdef abc abcd", ["abcd", "def", "abc"], [U, 2, 0]),
    ("djzhoyuhmcijprfwrssuhvgzw",
     ["##c", "d", "##d##f", "##g", "##h", "##hv", "##i", "##j", "##m", "##o", "##p", "##r", "##s",
      "##u", "##uh", "##w", "##y", "##z"], None),
    # testUtf8  tests.cpp:208-217
    ("привет мир", ["привет", "мир"], [0, 1]),
    ("привет мир", ["при", "##вет", "мир"], [0, 1, 2]),
    ("токенизация это круто", ["ток", "крут", "это", "##за", "##ция", "ция"], [U, 2, U]),
    ("токенизация это круто", ["ток", "крут", "это", "##за", "##ени", "##о", "##ция", "ция"],
     [0, 4, 3, 6, 2, 1, 5]),
]

# SURVEY.md §0.2 Q1..Q13 (text given as latin-1-escaped bytes where not valid UTF-8)
survey = [
    ("Q1", "ab-cd", ["ab-cd", "ab", "-", "cd"], [0]),
    ("Q2", "ab中cd", ["ab中cd", "ab", "中", "cd"], [0]),
    ("Q3", "中文", ["中", "文", "##文"], [0, 1]),
    ("Q4", b"a\xffb", ["ab", "a", "##b"], [0]),
    ("Q5a", "zz aa", ["aa", "[UNK]"], [1, 0]),
    ("Q5b", "zz aa", ["aa"], [U, 0]),
    ("Q6a", "", ["a"], []),
    ("Q6b", "  \n\t ", ["a"], []),
    ("Q7", "[CLS] a", ["[CLS]", "a", "[", "]", "CLS"], [2, 4, 3, 1]),
    ("Q8", "... a", ["...", ".", "a"], [1, 1, 1, 2]),
    ("Q9", "ab ab", ["ab", "x", "ab"], [0, 0]),
    ("Q10", "xaa", ["x", "##aa", "##a", "aa"], [0, 1]),
    ("Q11a", "a▁b", ["a", "b"], [0, 1]),
    ("Q11b", "a b", ["a", "b", "##b"], [U]),
    ("Q12", "abcz abc", ["ab", "##c", "abc"], [U, 2]),
    ("Q13", "a b", ["a b", "a", "b"], [0]),
]


def hexs(x):
    return (x if isinstance(x, bytes) else x.encode("utf8")).hex()


here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "reference_tests_cpp.json"), "w") as f:
    json.dump({"source": "/root/reference/tests/tests.cpp:137-217",
               "cases": [{"text_hex": hexs(t), "vocab_hex": [hexs(w) for w in v], "expected": e}
                         for t, v, e in tests_cpp]}, f, indent=1)
with open(os.path.join(here, "survey_probed_cases.json"), "w") as f:
    json.dump({"source": "SURVEY.md section 0.2 (outputs recorded from the reference public API)",
               "cases": [{"name": n, "text_hex": hexs(t), "vocab_hex": [hexs(w) for w in v],
                          "expected": e} for n, t, v, e in survey]}, f, indent=1)
print("wrote", len(tests_cpp), "+", len(survey), "cases")

"""JSON texts (written for this repository, in the style of the reference's technique blocks) and the paths queried in them.
tests/golden/make_golden.py runs them through the reference's vendored nlohmann::json 2.1.1 (oracle/_ref) and stores the
answers in tests/golden/json_pins.json; tests/test_oracle_pins.py asks the product's own reader the same questions."""

BLOCK = r'''
{
    "camera": { "origin": [ -4.5, 2.25, 1e1 ], "direction": [0, 1.5e-1, -0.0], "up": [0,1,0], "fovx": 60 },
    "objMesh": "conference/conference_exported.obj",
    "arealight": { "objMesh": "conference\/lights.obj", "intensity": [17, 12.5, 4, 0] },
    "photonfam": {
        "numLightPaths": 500000, "numVplLightPaths": 1024.0, "numMaxBounces": 3, "radiusPercentage": 3e-3,
        "numMaxIteration": -1, "timeLimitMs": 1.5E+4, "frameMode": "cleareveryframe", "misMode": "balance",
        "writeEveryFrame": false, "DoProgressive": true, "AlphaProgressive": 0.7, "clampingCoeff": 2.5,
        "big": 123456789012, "huge": 1e300, "tiny": 1e-320, "negfrac": -3.7, "posfrac": 3.7, "zero": 0, "negzero": -0.0,
        "exp": 1E3, "exp2": 25e-1, "intlike": 7.0, "maxint": 2147483647, "overint": 4294967295,
        "run": { "lightRender": true, "photonSplat": false, "vplSplat": true },
        "combinedFilename": "C:\\result\\conference_ours.pfm", "name": "tab\there \"quoted\" \u0041\u00e9\u20ac \ud83d\ude00 end",
        "empty": "", "nothing": null, "list": [], "obj": {}, "nested": [[1, 2], [3, [4, 5.5, "six", true, null]]]
    }
}
'''
BLOCK_PATHS = [
    "camera/origin/0", "camera/origin/1", "camera/origin/2", "camera/origin/3", "camera/origin", "camera/direction/1", "camera/direction/2",
    "camera/fovx", "camera/missing", "objMesh", "arealight/objMesh", "arealight/intensity/1", "arealight/intensity",
    "photonfam/numLightPaths", "photonfam/numVplLightPaths", "photonfam/numMaxBounces", "photonfam/radiusPercentage",
    "photonfam/numMaxIteration", "photonfam/timeLimitMs", "photonfam/frameMode", "photonfam/misMode", "photonfam/writeEveryFrame",
    "photonfam/DoProgressive", "photonfam/AlphaProgressive", "photonfam/clampingCoeff", "photonfam/big", "photonfam/huge", "photonfam/tiny",
    "photonfam/negfrac", "photonfam/posfrac", "photonfam/zero", "photonfam/negzero", "photonfam/exp", "photonfam/exp2", "photonfam/intlike",
    "photonfam/maxint", "photonfam/overint", "photonfam/run", "photonfam/run/lightRender", "photonfam/run/photonSplat", "photonfam/run/nope",
    "photonfam/combinedFilename", "photonfam/name", "photonfam/empty", "photonfam/nothing", "photonfam/list", "photonfam/list/0",
    "photonfam/obj", "photonfam/nested", "photonfam/nested/1/1/2", "photonfam/nested/1/1/3", "photonfam/nested/1/1/4", "photonfam/nested/1/1/1",
    "photonfam/nested/0/0", "photonfam", "pt", "",
]

CASES = [
    (BLOCK, BLOCK_PATHS),
    # whitespace, top-level scalars and arrays
    (" \t\r\n[ 1 ,\n2\t,3 ]\n ", ["", "0", "2", "3"]),
    ("42", [""]), ("-0", [""]), ('"just a string"', [""]), ("true", [""]), ("null", [""]), ("[]", ["", "0"]), ("{}", ["", "a"]),
    # duplicate keys
    ('{"a": 1, "a": 2, "b": {"c": 1, "c": [3]}}', ["a", "b/c", "b/c/0"]),
    # keys with escapes and empty keys
    ('{"": 5, "a\\/b": 6, "k\\u0041": 7, "sp ace": 8}', ["", "kA", "sp ace"]),
    # number grammar
    ("[1.0e+2, 1e-2, 0.5, 10, -10, 1E2, 0e0, 1.25e+0]", ["0", "1", "2", "3", "4", "5", "6", "7"]),
    # strings: every escape, 2- and 3-byte UTF-8 passed through unescaped
    ('["\\b\\f\\n\\r\\t\\"\\\\\\/", "caf\u00e9 \u20ac", "\\u00e9\\u20ac", "\\ud834\\udd1e"]', ["0", "1", "2", "3"]),
    # things that are NOT JSON
    ("{'a': 1}", [""]), ("{a: 1}", [""]), ('{"a": 1,}', ["a"]), ("[1, 2,]", ["0"]), ("[1 2]", ["0"]), ('{"a" 1}', ["a"]), ('{"a": }', ["a"]),
    ("01", [""]), ("+1", [""]), (".5", [""]), ("1.", [""]), ("1.e3", [""]), ("1e", [""]), ("-", [""]), ("0x10", [""]), ("NaN", [""]), ("nan", [""]),
    ("Infinity", [""]), ("inf", [""]), ("-inf", [""]), ("True", [""]), ("nul", [""]), ("", [""]), ("   ", [""]),
    ('{"a": 1} x', ["a"]), ('{"a": 1} {"b": 2}', ["a"]), ("[1] // comment", ["0"]), ("/* c */ [1]", ["0"]), ('"unterminated', [""]),
    ('"bad \\x escape"', [""]), ('"bad \\u12 escape"', [""]), ('"lone \\ud800 surrogate"', [""]), ('"tab\tinside"', [""]), ('"newline\ninside"', [""]),
    ('{"a": [1, {"b": [2, {"c": 3}]}', ["a"]), ("[[[[[[[[[[1]]]]]]]]]]", ["0/0/0/0/0/0/0/0/0/0"]),
]

"""Host-side parsers under AddressSanitizer + UndefinedBehaviorSanitizer (CPU tier): tests/cpp/fuzz_host.cpp runs the log readers
(LogReader, MappedLog incl. its bisection seek), the run-time .lcm schema (parse, value-tree decode, Plan::run / ::layout with a warm
Shape), the compiled-in pronto types, ModelClient::fromURDFString and the two replayers (SegmentBatcher::run, SegmentStreamer::run)
over deterministic mutations of valid inputs -- truncations, bit flips, splices, length-field edits.

The C ABI behind the replayers is replaced by HOST stubs generated here from include/pronto_batch.h (every entry point returns PB_OK;
allocations are heap blocks, uploads are memcpy), so the sanitizers also see every byte the replayers write into their chunk buffers
and every size they hand to an upload.  Test infrastructure only: nothing of it is linked into the product."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pronto_batch.h")

SPECIAL = {
    "pb_create": """{ (void) device; (void) n_snapshots; if (!out) return 1; int *c = (int *) calloc(4, sizeof(int)); c[0] = n_states; c[1] = batch; *out = (pb_ctx *) c; return 0; }""",
    "pb_destroy": """{ free(ctx); return 0; }""",
    "pb_last_error": """{ (void) ctx; return "stub"; }""",
    "pb_hot_kernel": """{ (void) ctx; return "stub"; }""",
    "pb_version": """{ return "host stub"; }""",
    "pb_batch": """{ return ((const int *) ctx)[1]; }""",
    "pb_n_states": """{ return ((const int *) ctx)[0]; }""",
    "pb_malloc": """{ (void) ctx; *dev_ptr = malloc(bytes ? bytes : 1); return *dev_ptr ? 0 : 2; }""",
    "pb_free": """{ (void) ctx; free(dev_ptr); return 0; }""",
    "pb_host_alloc": """{ (void) ctx; *host_ptr = malloc(bytes ? bytes : 1); return *host_ptr ? 0 : 2; }""",
    "pb_host_free": """{ (void) ctx; free(host_ptr); return 0; }""",
    "pb_memcpy_h2d": """{ (void) ctx; if (bytes) memcpy(dev_dst, host_src, bytes); return 0; }""",
    "pb_memcpy_d2h": """{ (void) ctx; if (bytes) memcpy(host_dst, dev_src, bytes); return 0; }""",
    "pb_upload_async": """{ (void) ctx; (void) after_fence; if (bytes) memcpy(dev_dst, host_src, bytes); return 0; }""",
    "pb_fence_create": """{ (void) ctx; static int next = 0; *fence_out = next++ % 16; return 0; }""",
    "pb_head_slot": """{ (void) ctx; return -1; }""",
    "pb_get_head": """{ const int n = ((const int *) ctx)[0]; (void) first; (void) mem;
  if (vec_out) { memset(vec_out, 0, sizeof(double) * n * count); }
  if (quat_out) { memset(quat_out, 0, sizeof(double) * 4 * count); }
  if (cov_out) { memset(cov_out, 0, sizeof(double) * n * n * count); }
  if (ll_out) { memset(ll_out, 0, sizeof(double) * count); }
  return 0; }""",
}


def make_stubs(path):
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = ['#include <cstdlib>\n#include <cstring>\n#include "%s"\nextern "C" {\n' % HEADER]
    n = 0
    for m in re.finditer(r"\n(int|const char \*)\s*(pb_\w+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        body = SPECIAL.get(name)
        if body is None:
            names = []
            for a in args.split(","):
                a = a.strip()
                if a in ("void", ""):
                    continue
                nm = re.sub(r"\[[^\]]*\]", "", a).split()[-1].lstrip("*")
                names.append(nm)
            body = "{ %s return 0; }" % " ".join("(void) %s;" % x for x in names)
        out.append("%s %s(%s) %s\n" % (ret, name, args, body))
        n += 1
    out.append("}\n")
    with open(path, "w") as f:
        f.write("".join(out))
    return n


def test_parsers_and_replayers_survive_damaged_inputs_under_asan_ubsan(tmp_path):
    build = os.path.join(ROOT, "tests", "build")
    os.makedirs(build, exist_ok=True)
    stubs = os.path.join(build, "fuzz_abi_stubs.cpp")
    assert make_stubs(stubs) >= 70          # every entry point of the header has a host stand-in
    exe = os.path.join(build, "fuzz_host")
    src = os.path.join(ROOT, "tests", "cpp", "fuzz_host.cpp")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-Wall", "-o", exe, src, stubs]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe, str(tmp_path), "300"], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-3000:], r.stderr[-6000:])
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr

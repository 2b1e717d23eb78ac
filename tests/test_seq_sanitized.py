"""CPU AddressSanitizer + UndefinedBehaviorSanitizer build of the host sequencer (vi_ekf_amd/csrc/viekf_seq.cpp + viekf_yaml.cpp)
against a TEST-ONLY host stub of the viekf_batch_* entry points (tests/cpp/seq_host_stub.cpp: not a filter, never shipped), driven
through the rewind / replay / ring wrap-around / queue-trim / keyframe / log-writer / independent-clock scenarios of
tests/test_gpu_sequencer.py (tests/cpp/seq_sanitized_driver.cpp).  GPU sanitizers are not available on the pool; the sequencer is
host code, so this is where its deque and ring indexing gets checked.  Reference plumbing restated: src/vi_ekf/vi_ekf_meas.cpp:6-194."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vi_ekf_amd", "csrc")
YAML = os.path.join(ROOT, "vi_ekf_amd", "params", "ekf.yaml")


def test_sequencer_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "seq_sanitized")
    # viekf_params_load_yaml lives in viekf_capi.hip (HIP); the driver only needs the YAML reader: a two-line shim over viekf_yaml.cpp
    shim = tmp_path / "yaml_shim.cpp"
    shim.write_text(r'''
#include <cstring>
#include <string>
#include "%s/viekf_host.hpp"
#include "%s/../../include/viekf.h"
using namespace viekf;
extern "C" int viekf_params_load_yaml(const char* path, viekf_params* p) {
  std::memset(p, 0, sizeof(*p));
  YamlMap m; std::string err, name; double v = 0;
  if (!yaml_parse_file(path, m, err)) return VIEKF_ERR_YAML;
  yaml_get_string(m, "name", name, err);
  yaml_get_doubles(m, "min_depth", &p->min_depth, 1, err);
  yaml_get_doubles(m, "keyframe_overlap_threshold", &p->keyframe_overlap_threshold, 1, err);
  yaml_get_doubles(m, "use_keyframe_reset", &v, 1, err); p->use_keyframe_reset = v != 0;
  yaml_get_doubles(m, "q_b_u", p->q_b_u, 4, err);
  yaml_get_doubles(m, "x0", p->x0, 17, err);
  yaml_get_doubles(m, "P0", p->P0, 16, err);
  yaml_get_doubles(m, "Qx", p->Qx, 16, err);
  yaml_get_doubles(m, "Qu", p->Qu, 6, err);
  yaml_get_doubles(m, "lambda", p->lambda, 16, err);
  yaml_get_doubles(m, "P0_feat", p->P0_feat, 3, err);
  yaml_get_doubles(m, "Qx_feat", p->Qx_feat, 3, err);
  yaml_get_doubles(m, "lambda_feat", p->lambda_feat, 3, err);
  yaml_get_doubles(m, "q_b_c", p->q_b_c, 4, err);
  yaml_get_doubles(m, "p_b_c", p->p_b_c, 3, err);
  p->use_partial_update = 1; p->use_drag_term = 1;
  return VIEKF_OK;
}
''' % (CSRC, CSRC))
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-pthread",
           "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-misleading-indentation", "-I" + os.path.join(ROOT, "include"), "-o", exe,
           os.path.join(ROOT, "tests", "cpp", "seq_sanitized_driver.cpp"), os.path.join(ROOT, "tests", "cpp", "seq_host_stub.cpp"),
           os.path.join(CSRC, "viekf_seq.cpp"), os.path.join(CSRC, "viekf_yaml.cpp"), str(shim)]
    subprocess.check_call(cmd)
    logs = tmp_path / "logs"
    logs.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, YAML, str(logs)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "sanitized sequencer scenarios: ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr

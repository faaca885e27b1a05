"""Every entry point of the C ABI called with a null context / null pointers: an error code comes back
(MCULL_E_INVALID, a zero count or a null pointer), nothing dereferences the null.  No GPU is needed: the
checks come before any device work.  Runs in a child process so that a crash would fail one test, not
the session."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r'''
import ctypes as C
import importlib
import sys

sys.path.insert(0, sys.argv[1])
mc = importlib.import_module("vulkan-renderer_amd")
lib = mc.lib
buf = (C.c_uint8 * 4096)()
P = C.cast(buf, C.c_void_p)
calls = {
 "mcull_destroy": (None,), "mcull_synchronize": (None,), "mcull_upload_meshlets": (None, P, 1, 0),
 "mcull_upload_bounds_soa": (None, P, P, 1, 0, 0), "mcull_upload_geometry": (None, P, 1, P, 1, 0),
 "mcull_compute_bounds": (None,), "mcull_download_meshlets": (None, P, 1), "mcull_upload_transforms": (None, P, 1, 0),
 "mcull_set_instance_ranges": (None, P, P, 1), "mcull_hiz_build": (None, 0, P, 4, 4, 0), "mcull_hiz_build_views": (None, 1, P, P, 4, 4, 0),
 "mcull_hiz_pyramid": (None, 0, None, None), "mcull_hiz_download_level": (None, 0, 0, P),
 "mcull_cull": (None, P, None, None), "mcull_cull_static": (None, P, None, None), "mcull_cull_views": (None, 1, P, None, None), "mcull_select_result": (None, 0),
 "mcull_cull_device_globals": (None, P, 1, None, None), "mcull_cull_views_device_globals": (None, 1, P, 1, None, None),
 "mcull_rebase_commands": (None, P, 0), "mcull_rebase_gathered_commands": (None, None, 1, 1, None, None), "mcull_task_payload_count": (None,), "mcull_emit_task_payloads": (None, P),
 "mcull_download_count": (None, None), "mcull_download_list": (None, P, 1, None), "mcull_download_commands": (None, P, 8),
 "mcull_download_mask": (None, P, 1), "mcull_precull_stats": (None, None, None, None), "mcull_precull_group_stats": (None, None, None), "mcull_slow_path_pairs": (None, None), "mcull_enable_timing": (None, 1), "mcull_last_kernel_ms": (None, None),
 "mcull_depth_rasterize": (None, P, 4, 4, 1), "mcull_depth_rasterize_device_globals": (None, P, 4, 4, 1),
 "mcull_depth_buffer": (None, None, None, None), "mcull_depth_download": (None, P), "mcull_scene_upload": (None, b"/nonexistent"),
 "mcull_mesh_upload": (None, None), "mcull_mesh_build_meshlets_gpu": (None, None),
 "mcull_mesh_from_arrays_gpu": (None, P, 1, None, 0, None, 0, P, 3, None),
 "mcull_group_destroy": (None,), "mcull_group_size": (None,), "mcull_group_ctx": (None, 0), "mcull_group_cull": (None, P, None, None),
 "mcull_group_synchronize": (None,), "mcull_group_frame": (None, P, 4, 4, 3, 0, None, None), "mcull_group_download": (None, 0, None, 0, None, None),
 "mcull_mesh_optimize": (None,), "mcull_mesh_build_meshlets": (None,), "mcull_mesh_counts": (None, None), "mcull_mesh_save": (None, b"/tmp/x"),
 "mcull_mesh_vertices": (None,), "mcull_mesh_indices": (None,), "mcull_mesh_meshlet_data": (None,), "mcull_mesh_meshlets": (None,),
 "mcull_create": (0, None, None), "mcull_group_create": (None, 1, None),
 "mcull_create_globals": (None, None, None, 45.0, 0.1, 800, 600, 1, 1, None), "mcull_extract_frustum": (None,),
 "mcull_reference_transforms": (1, 42, None), "mcull_create_indirect_buffer": (1, 1, 1, None),
 "mcull_scene_info": (None, None), "mcull_scene_load": (None, None, None, None, None, None, None, None),
 "mcull_scene_save": (None, None, 0, None, 0, None, 0, None, 0, None, None, 0), "mcull_mesh_load_obj": (None, None),
 "mcull_mesh_from_arrays": (None, 0, None, 0, None, 0, None, 0, None),
 "mcull_rccl_unique_id": (None,), "mcull_rccl_comm_create": (None, 1, 0, 0, None),
 "mcull_allgather_commands": (None, None, None, None, 1, 1, None, None),
}

# entry points without a pointer that could be null, or that accept null by contract
exempt = {"mcull_last_error", "mcull_status_string", "mcull_version", "mcull_hiz_level_count", "mcull_hiz_layout",
          "mcull_mesh_destroy", "mcull_group_last_error", "mcull_collective_last_error", "mcull_rccl_comm_destroy"}
missing = set(mc.ABI_SYMBOLS) - set(calls) - exempt
assert not missing, missing
for name, args in calls.items():
    r = getattr(lib, name)(*args)
    ok = r is None or r == 0 if name in ("mcull_task_payload_count", "mcull_group_size", "mcull_group_ctx", "mcull_mesh_vertices",
                                          "mcull_mesh_indices", "mcull_mesh_meshlet_data", "mcull_mesh_meshlets") else r == mc.E_INVALID
    assert ok, (name, r)
assert lib.mcull_last_error(None) == b"null context" and lib.mcull_group_last_error(None) == b"null group"
lib.mcull_mesh_destroy(None)
assert lib.mcull_rccl_comm_destroy(None) == 0 and lib.mcull_collective_last_error() is not None
print("probed", len(calls))
'''


def test_null_arguments_are_rejected_not_dereferenced(tmp_path):
    script = tmp_path / "probe.py"
    script.write_text(PROBE)
    out = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "probed" in out.stdout

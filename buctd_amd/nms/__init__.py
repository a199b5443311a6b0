"""lib/nms counterpart: import from buctd_amd.nms.nms like the reference imports from nms.nms."""

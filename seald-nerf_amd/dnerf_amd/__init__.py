"""Host-side driver of the MI355X dynamic-NeRF rendering path: synthetic scenes, the field network
and the render loop that the reference keeps in dnerf/network.py and dnerf/renderer.py."""

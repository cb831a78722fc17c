"""Drop-in for the reference's external CUDA module `simple_knn` (setup.sh:35-37): the only symbol the
reference uses is `simple_knn._C.distCUDA2` (scene/gaussian_model.py:20, scene/mesh_gaussian_model.py:23)."""

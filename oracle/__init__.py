"""TEST INFRASTRUCTURE ONLY — CPU oracle for the MI355X denoising hot path.

Nothing under ``multimodal_diffusion_amd/`` may import this package.  The only
legal importers are ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` (as the checker / reported baseline, never
as the thing that is shipped or measured as the product).
"""
